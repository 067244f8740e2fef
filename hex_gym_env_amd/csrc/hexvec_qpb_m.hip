// lanes-per-board bitboard kernels (hexvec_qpb.h), board sizes of this translation unit
#define HV_QPB_SIZES(X) X(9) X(10) X(11)
#define HV_QPB_NAME(x) qpb_m_##x
#include "hexvec_qpb.h"
