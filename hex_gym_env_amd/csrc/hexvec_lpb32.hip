// lane-per-board kernels, 32 boards per wavefront (see hexvec_lpb.h)
#define HV_BPW 32
#undef HV_HELPER
#define HV_HELPER 0 // (measured: the helper wave does not pay at 32 boards per wave)
#define HV_LPB_NAME(x) lpb32_##x
#include "hexvec_lpb.h"
