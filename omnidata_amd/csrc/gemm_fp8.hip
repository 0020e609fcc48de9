// gemm_fp8.hip -- the e4m3 (v_mfma_scale_f32_32x32x64_f8f6f4) instantiations of the implicit-GEMM kernels (gemm_impl.h).
#include "gemm_impl.h"

namespace dptx {

hipError_t launch_gemm_fp8(const GemmParams& p, hipStream_t stream) { return launch_dt<DT_FP8, 1>(p, stream); }

}  // namespace dptx
