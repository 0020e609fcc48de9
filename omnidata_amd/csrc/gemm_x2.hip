// gemm_x2.hip -- the 2-MFMA form of the fp16 hi/lo-plane kernels (gemm_impl.h XT == 2, GemmParams::a_hi_only): the weights
// contribute both planes, the activations their hi plane only -- per-layer precision 2 of the "mixed" dtype.
#include "gemm_impl.h"

namespace dptx {

hipError_t launch_gemm_x2(const GemmParams& p, hipStream_t stream) { return launch_dt<DT_FP16, 2, 2, 2>(p, stream); }

}  // namespace dptx
