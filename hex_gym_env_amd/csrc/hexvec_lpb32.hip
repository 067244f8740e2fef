// lane-per-board kernels, 32 boards per wavefront (see hexvec_lpb.h)
#define HV_BPW 32
#define HV_LPB_NAME(x) lpb32_##x
#include "hexvec_lpb.h"
