/* FIVE.h -- forwarder: the drop-in API lives in frirl_dropin.h (reference header of the same name: src/five/FIVE.h). */
#include "frirl_dropin.h"
