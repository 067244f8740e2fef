// lanes-per-board bitboard kernels (hexvec_qpb.h), board sizes of this translation unit
#define HV_QPB_SIZES(X) X(4) X(5) X(6) X(7) X(8)
#define HV_QPB_NAME(x) qpb_s_##x
#include "hexvec_qpb.h"
