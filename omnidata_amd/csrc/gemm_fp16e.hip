// gemm_fp16e.hip -- single-plane fp16 compute with the TWO-plane epilogue (per-layer precision policy of the "mixed" dtype:
// a layer that spends one MFMA per product can still read hi/lo residuals and write a hi/lo result; gemm_impl.h PLE).
#include "gemm_impl.h"

namespace dptx {

hipError_t launch_gemm_fp16e(const GemmParams& p, hipStream_t stream) { return launch_dt<DT_FP16, 1, 2>(p, stream); }

}  // namespace dptx
