// kernels.h -- host-visible launchers of the dptx HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace dptx {

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: sets it once per (kernel, current
// device) -- one process may hold engines on several GPUs (misc.hip)
void ensure_dyn_smem(const void* kernel, size_t bytes);

// One implicit-GEMM problem: C[M,N] = epilogue( gatherA[M,K] * W[N,K]^T ).
// A is addressed as an NHWC image batch so that a dense row-major matrix, a strided 1x1
// convolution and a kxk convolution are the same code path:
//   m -> (img, oy, ox) with a_rpi = Hout*Wout rows per image;  k -> (ky, kx, c), c < Cin
//   elem = A + img*a_img_stride + a_off + ((oy*stride+ky-pad_t)*Win + (ox*stride+kx-pad_l))*a_pix_stride + c
// out-of-image taps read as zero.  K = ksz*ksz*Cin, Cin % 64 == 0, K % 64 == 0, N % BN == 0.
struct GemmParams {
  const void* A;
  const void* W;  // [N][K] 16-bit, K contiguous, K ordered (ky,kx,c)
  void* C;
  const float* bias;  // [N] or (bias_per_img) [imgs][N]; may be null
  const void* R1;     // residual, same addressing as C; may be null
  const void* R2;     // second residual; may be null
  int M, N, K;
  int ldw;  // elements between rows of W (>= K; ProjectReadout uses half of a [768][1536] matrix)
  int a_rpi, Wout, Hin, Win, Cin, a_pix_stride;
  long long a_img_stride, a_off;
  long long a_bytes;  // size of the A buffer in bytes (bounds of the buffer descriptor; glds path needs < 2^31)
  int ksz, stride, pad_t, pad_l;
  // C row of m: img = m / c_rpi, p = m % c_rpi -> row = img*c_img_rows + c_row_off + p
  int c_rpi, c_img_rows, c_row_off, ldc;
  int r2_bcast;      // R2 row = c_row_off + p (broadcast over images: pos_embed)
  int bias_per_img;  // bias row = img (of the C mapping)
  int act;           // 0 none, 1 relu, 2 gelu(erf)
  int a_relu;        // relu applied to A while staging (RCU pre-activation)
  int a_fp32;        // A elements are fp32 (converted while staging)
  int c_fp32, r1_fp32, r2_fp32;
  Planes planes;  // hi->lo plane distances (bf16x3 mode only)
  // per-layer precision policy (dtype "mixed"): epi2 = 1 runs the two-plane epilogue on a single-pass (one MFMA per product)
  // fp16 kernel -- residuals are read with their lo planes and C is written as a hi/lo pair; *_hi_only (two-plane epilogues,
  // also of the 3-MFMA kernels) switch off the lo plane of C / R1 / R2 individually
  int epi2, c_hi_only, r1_hi_only, r2_hi_only;
  // a_hi_only (fp16 hi/lo-plane kernels): TWO MFMAs per product -- a_hi w_hi + a_hi w_lo; the lo plane of A is not read
  // (per-layer precision 2 of dptx_set_layer_precision: the weights keep their full precision, the activations are rounded
  // to fp16 once)
  int a_hi_only;
  int xcd_m, xcd_n;  // XCD grid of the tile partition (filled in by launch_gemm)
  float cu_share, cu_share_small;  // part of the chip this launch can count on, for the 256x256 rule / the narrow-tile
                                   // thresholds (filled in by launch_gemm from gemm_set_cu_share; 0 = 1)
  int k_tap_fast;    // visit the k-tiles taps-fastest inside a 64-channel block (3x3 convs with Cin >= 512: +12..17 %)
  // GroupNorm(32) statistics of the output from the epilogue (null: none): partial[img][gn_blocks][32] float2 records
  // of (sum, sum of squares) per 32-row block; gn_hw = rows per image (% 32 == 0), gn_cpg = N / 32 channels per group
  float* gn_part;
  int gn_hw, gn_blocks, gn_cpg;
  // fp8 (MODE_FP8: A and W are e4m3 bytes, K and ldw count bytes): C *= out_scale * out_scale_v[n] before the bias
  // (out_scale 0 = none; out_scale_v: optional per-output-channel vector -- the inverse of the power-of-two scale the
  // weights of channel n were quantised with; out_scale then carries the inverse of the activation scale).  C8: optional
  // e4m3 copy of the final output times q_scale (0 = 1.0; same addressing as C, 1 byte per element), ReLU'd first when
  // q_relu -- also available to the 16-bit modes (the producers of an fp8 GEMM's input run in bf16)
  void* C8;
  int q_relu;
  float out_scale, q_scale;
  const float* out_scale_v;
  // LayerNorm folded into the GEMMs around it (timm Block: x += proj(attn(norm1(x))); x += fc2(gelu(fc1(norm2(x))))):
  //  * PRODUCER side (patch-embed, proj, fc2 -- the launches that write the fp32 token stream): C16 = 16-bit copy of the
  //    final output rows (the next GEMM's A operand), row_stats = per (C row, 128-column block) float2 (sum, sum of
  //    squares) of the fp32 output, [rows][stats_nblk = 8] (row stride; N / 128 of them are written); needs
  //    N % 128 == 0, N <= 1024 and a tile at least 128 columns wide.
  //  * CONSUMER side (qkv, fc1): A is that 16-bit copy of the UN-normalised stream, W = gamma (.) W and bias = W beta + b
  //    were folded at pack time, ln_colsum[n] = sum_k W'[n][k] (of the ROUNDED operand values, so that the mean
  //    component cancels exactly), and the epilogue applies  y = (acc - mu * colsum) * rstd + bias  with (mu, rstd) of
  //    GEMM row m combined from the records ln_stats[m][0..ln_nblk) (ln_nblk = 6 or 8, row stride 8).
  void* C16;
  float* row_stats;
  int stats_nblk;
  const float* ln_stats;
  const float* ln_colsum;
  int ln_nblk;
  float ln_eps, ln_inv_dim;
  long long* trace;  // debug: s_memtime stamps of block 0 (dptx_debug_set_trace); null in production
  int debug_flags;   // debug (dptx_debug_set_gemm_flags): 1 = staged epilogue everywhere, 2 = one block per tile, 4 = lockstep two-plane 128x128 kernel
  float a_rpi_rcp, wout_rcp;  // 1 / a_rpi, 1 / Wout (filled in by launch_gemm: row -> (image, y, x) without integer division)
};

// Fills the "plain dense row-major" defaults for A [M,K] (lda = K) and C [M,N].
void gemm_params_dense(GemmParams& p, int M, int N, int K);
// dtype 0 bf16 / 1 fp16.  Picks the tile configuration from (M, N).  Returns hipError_t.
hipError_t launch_gemm(int dtype, const GemmParams& p, hipStream_t stream);
// debug: every following GEMM launch stamps s_memtime per k-tile phase for the 8 waves of block 0 into dev_buf
// ([wave][64 k-tiles][4] int64; null switches it off)
void gemm_set_trace(long long* dev_buf);
// debug / tests: 1 = no register-direct epilogue, 2 = no persistent launch (same results either way)
void gemm_set_debug_flags(int flags);
// tile selection: the following launches share the chip with (1 / share - 1) concurrent streams of the same forward
void gemm_set_cu_share(float share, float share_small = 0.f);

hipError_t launch_attention(int mode, const void* qkv, void* out, int B, int S, int heads, Planes pl, hipStream_t stream);

hipError_t launch_layernorm(int mode, const float* x, const float* gamma, const float* beta, void* y,
                            int M, int C, float eps, Planes pl, hipStream_t stream);

// GroupNorm(32) on NHWC 16-bit: stats pass writes partial[B][chunks][32][2] fp32,
// apply pass: Y = act( gn(X) + (R ? (r_gamma ? gn2(R) : R) : 0) ).  In-place (Y==X) allowed.
struct GnParams {
  const void* X; void* Y;
  const float* gamma; const float* beta;  // [C]
  float* partial;                         // [B][chunks][32][2]
  const void* R;                          // optional residual (same shape)
  const float* r_gamma; const float* r_beta; const float* r_partial;  // optional GN on residual
  int B, HW, C, relu;
  float eps;
  int nrec;  // partial records per image in `partial` / `r_partial`: 0 = gn_chunks(HW, C) (written by launch_gn_stats),
             // HW / 32 when a GEMM epilogue wrote them (GemmParams::gn_part)
};
int gn_chunks(int HW, int C);  // blocks per image of the GroupNorm stats/apply grids = partial records per image
hipError_t launch_gn_stats(int mode, const void* X, float* partial, int B, int HW, int C, Planes pl, hipStream_t stream);
hipError_t launch_gn_apply(int mode, const GnParams& p, Planes pl, hipStream_t stream);
// stem: Y[B,Ho,Wo,C] = maxpool3x3s2_SAME( relu(gn(X[B,H,W,C])) )
hipError_t launch_gn_relu_maxpool(int mode, const void* X, void* Y, const float* gamma, const float* beta,
                                  const float* partial, int B, int H, int W, int C, float eps, Planes pl, hipStream_t stream);

// fused stem conv 7x7 s2 TF-SAME: x NCHW fp32 [B,3,H,W] -> y NHWC 16-bit [B,H/2,W/2,64]; Wt [64][176], k = (c*7+ky)*8 + kx
// io: element type of x (common.h IO_*: fp32 / bf16 / fp16)
hipError_t launch_stem_conv(int mode, const void* x, int io, const void* Wt, void* y, int B, int H, int W, Planes pl,
                            hipStream_t stream);

// bilinear x2 align_corners=True on NHWC 16-bit
// Y8 (optional): e4m3 copy of the output times q_scale, 1 byte per element, for an fp8 convolution downstream
hipError_t launch_upsample2x(int mode, const void* X, void* Y, int B, int H, int W, int C, Planes pl, hipStream_t stream,
                             void* Y8 = nullptr, float q_scale = 1.0f, bool out_hi_only = false);
// fp8 calibration: atomicMax(*amax_bits, bits of max |x| (relu: max(x, 0)) over n 16-bit elements); *amax_bits starts at 0
hipError_t launch_amax(int mode, const void* X, size_t n, int relu, unsigned* amax_bits, hipStream_t stream);

// range check: *flag |= 1 if the 16-bit tensor X (n elements, n % 8 == 0; hi plane only) holds an Inf or a NaN
hipError_t launch_nonfinite_scan(int mode, const void* X, size_t n, unsigned* flag, hipStream_t stream);

// debug: *out += 64-bit sum of the 32-bit words of [p, p + bytes)
hipError_t launch_checksum(const void* p, size_t bytes, unsigned long long* out, hipStream_t stream);
hipError_t launch_spin(long long ticks_100mhz, hipStream_t stream);   // one sleeping wave (dptx_probe_stream_overlap)

// y NCHW fp32 [B,Cout,HW] = act( W[Cout][32] * x[B*HW,32] + b ),  Cout <= 4
hipError_t launch_head_out(int mode, const void* X, const float* w, const float* b, void* y, int io, int B, int HW,
                           int Cout, int relu, Planes pl, hipStream_t stream);

// X[b*577] = cls + pos[0]  (fp32 token stream)
// fused tail of the head: x2 bilinear -> conv3x3 128->32 -> ReLU -> conv1x1 32->C -> ReLU (head.hip)
// (hi / lo-plane modes: head_tail_x3_kernel, three MFMAs per product; `pl` = the plane distances of H0 and W2)
hipError_t launch_head_tail(int mode, const void* H0, const void* W2, const float* b2, const float* w4, const float* b4,
                            void* y, int io, int B, int Hs, int Ws, int C, int relu_out, hipStream_t stream, Planes pl = Planes{0, 0});
hipError_t launch_pos_resize(const float* src, float* dst, int g_old, int gh, int gw, int C, hipStream_t stream);
// X16 / stats (optional, LayerNorm fold): 16-bit copy of the rows and their (sum, sum of squares) per 128-column block; X may
// be null (16-bit token stream: no fp32 copy)
hipError_t launch_cls_rows(int mode, const float* cls, const float* pos, float* X, int B, int S, int C, void* X16, float* stats,
                           hipStream_t stream, void* X8 = nullptr, float q_scale = 1.0f);

// out[b][n] = bias[n] + sum_k x[b*x_stride + k] * W[n*ldw + w_off + k]   (x fp32, W 16-bit, out fp32)
// x16 != 0: x points at 16-bit values (the 16-bit token stream of the single-pass dtypes)
hipError_t launch_readout_cls(int mode, const float* x, long long x_stride, const void* W, int ldw, int w_off,
                              const float* bias, float* out, int B, int N, int K, Planes pl, hipStream_t stream, int x16 = 0);
// fp32 -> 16-bit (hi/lo planes in bf16x3 mode), n % 8 == 0
hipError_t launch_cast_f32(int mode, const float* src, void* dst, size_t n, Planes pl, hipStream_t stream);

// DPT-Large: x NCHW [B,3,H,W] (io type) -> 16x16 patches P[B*(H/16)*(W/16)][768], k = (c, ky, kx)
hipError_t launch_patchify16(int mode, const void* x, int io, void* P, int B, int H, int W, Planes pl, hipStream_t stream);
// G[(b,y,x)][(dy*k+dx)*C + c] -> Y[b][y*k+dy][x*k+dx][c]  (the re-layout half of a ConvTranspose2d with kernel == stride)
hipError_t launch_depth_to_space(int mode, const void* G, void* Y, int B, int h, int w, int k, int C, Planes pl, hipStream_t stream);

// generic 16-bit / fp32 -> fp32 copy for taps
hipError_t launch_to_f32(int mode, const void* src, float* dst, size_t n, Planes pl, hipStream_t stream);

}  // namespace dptx
