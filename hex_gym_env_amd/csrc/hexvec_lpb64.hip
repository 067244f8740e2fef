// lane-per-board kernels, 64 boards per wavefront (see hexvec_lpb.h)
#define HV_BPW 64
#define HV_LPB_NAME(x) lpb64_##x
#include "hexvec_lpb.h"
