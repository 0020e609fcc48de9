// gemm.hip -- launch_gemm and the single-plane bf16 instantiations of the implicit-GEMM kernels (gemm_impl.h).
#include "gemm_impl.h"

namespace dptx {

hipError_t launch_gemm_16(int dt, const GemmParams& p, hipStream_t stream) {
  return dt == DT_BF16 ? launch_dt<DT_BF16, 1>(p, stream) : launch_gemm_fp16(p, stream);
}

void gemm_params_dense(GemmParams& p, int M, int N, int K) {
  p = GemmParams{};
  p.M = M; p.N = N; p.K = K; p.ldw = K;
  p.a_rpi = M > 0 ? M : 1; p.Wout = p.a_rpi; p.Hin = 1; p.Win = p.a_rpi; p.Cin = K; p.a_pix_stride = K;
  p.a_img_stride = 0; p.a_off = 0; p.ksz = 1; p.stride = 1; p.pad_t = 0; p.pad_l = 0;
  p.c_rpi = 0x7fffffff; p.c_img_rows = 0; p.c_row_off = 0; p.ldc = N;
  p.a_bytes = (long long)M * K * 2;
}

// debug state is per host thread, like the tile-share state below: a test thread that switches launch forms or records a
// trace does not change what another thread's forwards launch
static thread_local long long* g_trace = nullptr;
void gemm_set_trace(long long* dev_buf) { g_trace = dev_buf; }
static thread_local int g_debug_flags = 0;
void gemm_set_debug_flags(int flags) { g_debug_flags = flags; }
static thread_local float g_cu_share = 1.0f, g_cu_share_small = 1.0f;
void gemm_set_cu_share(float share, float share_small) {
  g_cu_share = share > 0.f && share <= 1.f ? share : 1.0f;
  g_cu_share_small = share_small > 0.f && share_small <= 1.f ? share_small : g_cu_share;
}

hipError_t launch_gemm(int mode, const GemmParams& p0, hipStream_t stream) {
  GemmParams p = p0;
  p.trace = g_trace;
  p.debug_flags = g_debug_flags;
  if (p.cu_share <= 0.f) { p.cu_share = g_cu_share; p.cu_share_small = g_cu_share_small; }
  p.a_rpi_rcp = 1.0f / (float)(p.a_rpi > 0 ? p.a_rpi : 1);
  p.wout_rcp = 1.0f / (float)(p.Wout > 0 ? p.Wout : 1);
  if (p.gn_part != nullptr && (p.gn_hw % 32 != 0 || p.N % 32 != 0 || p.gn_cpg != p.N / 32 || p.gn_cpg < 2 || p.gn_cpg > 32 ||
                               (p.gn_cpg & (p.gn_cpg - 1)) != 0 || p.bias != nullptr || p.act != 0))
    return hipErrorInvalidValue;  // statistics are those of the raw accumulators: no bias / activation in front of a GroupNorm
  // LayerNorm fold (kernels.h): producer side needs whole 128-column blocks per tile, consumer side a plain dense epilogue
  // (round 6: both sides of the fold also on the e4m3 kernel -- DPTX_FLAG_FP8_VIT; the fp8 output scale is applied first)
  if (p.row_stats != nullptr && (p.N % 128 != 0 || p.stats_nblk != 8 || p.N / 128 > 8)) return hipErrorInvalidValue;
  if (p.ln_stats != nullptr && (p.R1 != nullptr || p.R2 != nullptr || p.bias_per_img || p.ln_colsum == nullptr || (p.ln_nblk != 6 && p.ln_nblk != 8) ||
                                p.c_rpi != 0x7fffffff || p.gn_part != nullptr))
    return hipErrorInvalidValue;
  if (mode == MODE_FP8) {  // 128 e4m3 per k-tile row
    if (p.K % 128 != 0 || p.Cin % 128 != 0 || p.M <= 0 || p.N % 32 != 0 || p.ldw < p.K || p.ldw % 16 != 0 || p.gn_part != nullptr)
      return hipErrorInvalidValue;
    return launch_gemm_fp8(p, stream);
  }
  if (p.K % BK != 0 || p.Cin % BK != 0 || p.M <= 0 || p.N % 32 != 0 || p.ldw < p.K || p.ldw % 8 != 0) return hipErrorInvalidValue;
  if (mode == MODE_BF16) return p.epi2 ? hipErrorInvalidValue : launch_gemm_16(DT_BF16, p, stream);
  if (mode == MODE_FP16) return p.epi2 ? launch_gemm_fp16e(p, stream) : launch_gemm_16(DT_FP16, p, stream);
  if (mode == MODE_BF16X3) return launch_gemm_x3(DT_BF16, p, stream);
  if (mode == MODE_FP16X3) return p.a_hi_only ? launch_gemm_x2(p, stream) : launch_gemm_x3(DT_FP16, p, stream);
  return hipErrorInvalidValue;
}

}  // namespace dptx
