/* frirl_utils.h -- forwarder: the drop-in API lives in frirl_dropin.h (reference header of the same name: src/frirl/frirl_utils.h). */
#include "frirl_dropin.h"
