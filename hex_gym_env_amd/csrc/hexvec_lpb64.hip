// lane-per-board kernels, 64 boards per wavefront (see hexvec_lpb.h)
#define HV_BPW 64
#ifndef HV_HELPER
#define HV_HELPER 1 // helper wave for the observation copy (hexvec_lpb.h)
#endif
#define HV_LPB_NAME(x) lpb64_##x
#include "hexvec_lpb.h"
