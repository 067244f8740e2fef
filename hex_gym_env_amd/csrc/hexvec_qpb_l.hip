// lanes-per-board bitboard kernels (hexvec_qpb.h), board sizes of this translation unit
#define HV_QPB_SIZES(X) X(12) X(13) X(14)
#define HV_QPB_NAME(x) qpb_l_##x
#include "hexvec_qpb.h"
