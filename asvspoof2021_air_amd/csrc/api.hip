// Library identification for the C-ABI (include/air_hip.h).
#include "air_common.h"

extern "C" {
const char* air_version(void) { return "air_hip gfx950 1"; }
int air_abi_version(void) { return 1; }
}
