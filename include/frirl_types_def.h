/* frirl_types_def.h -- forwarder + the default descriptor every application copies
 * (`struct frirl_desc frirl = frirl_desc_default;`, reference src/frirl/frirl_types_def.h:22). */
#ifndef FRIRL_TYPES_DEF_H
#define FRIRL_TYPES_DEF_H
#include "frirl_dropin.h"
static const struct frirl_desc frirl_desc_default = FRIRL_DESC_DEFAULT_INITIALIZER;
#endif
