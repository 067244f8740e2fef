// lanes-per-board bitboard kernels (hexvec_qpb.h), board sizes of this translation unit
// (15x15 and 16x16 take 8 lanes per board, 17x17..19x19 sixteen: one DPP row)
#define HV_QPB_SIZES(X) X(15) X(16) X(17) X(18) X(19)
#define HV_QPB_NAME(x) qpb_x_##x
#include "hexvec_qpb.h"
