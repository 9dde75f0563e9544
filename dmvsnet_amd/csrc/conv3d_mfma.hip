// K3 translation unit -- filled in by the MFMA implicit-GEMM kernel (see git history of this round).
#include "common.h"
extern "C" int dmvs_conv3d_mfma(const float*, float*, const float*, const float*, const float*, const float*, int, int,
                                int, int, int, int, int, int, dmvs_stream_t) { return DMVS_EUNSUPPORTED; }
extern "C" long dmvs_conv3d_mfma_weight_floats(int, int, int, int) { return 0; }
extern "C" int dmvs_pack_conv_weights_mfma(const float*, float*, int, int, int, int) { return DMVS_EUNSUPPORTED; }
