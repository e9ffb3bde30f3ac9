/* frirl_test.h -- forwarder: the drop-in API lives in frirl_dropin.h (reference header of the same name: src/frirl/frirl_test.h). */
#include "frirl_dropin.h"
