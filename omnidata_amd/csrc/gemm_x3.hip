// gemm_x3.hip -- the hi/lo-plane (3 MFMAs per product) instantiations of the implicit-GEMM kernels (gemm_impl.h).
#include "gemm_impl.h"

namespace dptx {

hipError_t launch_gemm_x3(int dt, const GemmParams& p, hipStream_t stream) {
  return dt == DT_BF16 ? launch_dt<DT_BF16, 2>(p, stream) : launch_dt<DT_FP16, 2>(p, stream);
}

}  // namespace dptx
