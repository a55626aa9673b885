// common.hip -- library identification.
#include <hip/hip_runtime.h>
#include "savp_hip.h"

extern "C" const char* savp_version(void) { return "savp_hip 0.1 gfx950"; }
