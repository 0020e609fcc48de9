// gemm_fp16.hip -- the single-plane fp16 instantiations of the implicit-GEMM kernels (gemm_impl.h).
#include "gemm_impl.h"

namespace dptx {

hipError_t launch_gemm_fp16(const GemmParams& p, hipStream_t stream) { return launch_dt<DT_FP16, 1>(p, stream); }

}  // namespace dptx
