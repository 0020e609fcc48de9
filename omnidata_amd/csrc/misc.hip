// misc.hip -- the HBM-bound glue kernels of the DPT-Hybrid forward: bilinear x2
// (align_corners=True), the 32->C head projection writing NCHW fp32, cls-token rows, the cls
// half of ProjectReadout, and an fp32 export for debug taps.
#include "common.h"
#include "kernels.h"

#include <algorithm>
#include <mutex>
#include <set>
#include <utility>

namespace dptx {

void ensure_dyn_smem(const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::set<std::pair<const void*, int>> done;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  if (done.insert({kernel, dev}).second)
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// ------------------------------------------------------------------ bilinear x2, align_corners
// blocks.py:335-337 / dpt_depth.py:93: F.interpolate(scale_factor=2, mode="bilinear",
// align_corners=True).  Index/lambda arithmetic follows ATen's fp32 formulation:
// ratio = (in-1)/(out-1); src = ratio*dst; i0 = int(src); i1 = i0 + (i0 < in-1); l1 = src - i0.
// grid (ceil(Wo*C/8 / 256), B*Ho): the output row (b, oy) comes from blockIdx.y (one scalar division per
// block), a thread owns 8 channels of one output pixel; C/8 is a power of two on every call site, so the
// per-thread index math is shifts and masks.
template <int DT, int PL>
__global__ __launch_bounds__(256) void upsample2x_kernel(const uint16_t* __restrict__ X, uint16_t* __restrict__ Y,
                                                         uint8_t* __restrict__ Y8, float q_scale, int B, int H, int W, int C,
                                                         int cshift, long long plane, int out_hi_only) {
  const int Ho = 2 * H, Wo = 2 * W, cvec = C >> 3;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= Wo * cvec) return;
  const int v = idx & (cvec - 1), ox = idx >> cshift;
  const int b = blockIdx.y / Ho, oy = blockIdx.y - b * Ho;
  const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  const float rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  const float sy = ry * (float)oy, sx = rx * (float)ox;
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
  const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const uint16_t* img = X + (long long)b * H * W * C + v * 8;
  float a[8], bb[8], c[8], d[8], o[8];
  load8f<DT, PL>(img + (y0 * W + x0) * C, plane, a);
  load8f<DT, PL>(img + (y0 * W + x1) * C, plane, bb);
  load8f<DT, PL>(img + (y1 * W + x0) * C, plane, c);
  load8f<DT, PL>(img + (y1 * W + x1) * C, plane, d);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = bilerp(a[e], bb[e], c[e], d[e], lx0, lx1, ly0, ly1);
  // (out_hi_only: two planes in, the hi plane out -- the same rounding as the hi half of the pair)
  if (PL == 2 && out_hi_only) store8f<DT, 1>(Y + ((long long)blockIdx.y * Wo + ox) * C + v * 8, plane, o);
  else store8f<DT, PL>(Y + ((long long)blockIdx.y * Wo + ox) * C + v * 8, plane, o);
  if (Y8 != nullptr) {  // e4m3 copy for an fp8 conv (times the tensor's calibrated power-of-two scale)
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] *= q_scale;
    *(uint2*)(Y8 + ((long long)blockIdx.y * Wo + ox) * C + v * 8) = pack_fp8x8(o);
  }
}

hipError_t launch_upsample2x(int mode, const void* X, void* Y, int B, int H, int W, int C, Planes pl, hipStream_t stream,
                             void* Y8, float q_scale, bool out_hi_only) {
  const int cvec = C / 8;
  if (C % 8 != 0 || (cvec & (cvec - 1)) != 0 || (long long)H * W * C >= (1ll << 31)) return hipErrorInvalidValue;
  int cshift = 0;
  while ((1 << cshift) < cvec) ++cshift;
  dim3 grid((2 * W * cvec + 255) / 256, B * 2 * H);
  DPTX_DISPATCH_MODE(mode, hipLaunchKernelGGL((upsample2x_kernel<DT, PL>), grid, dim3(256), 0, stream, (const uint16_t*)X,
                                              (uint16_t*)Y, (uint8_t*)Y8, q_scale, B, H, W, C, cshift, pl.act, (int)out_hi_only));
  return hipGetLastError();
}

// ------------------------------------------------------------------------------ fp8 calibration
// max |x| (relu: max(x, 0)) of a 16-bit tensor; non-negative floats order like their bit patterns, so one atomicMax on
// the bits per block is the whole cross-block reduction
template <int DT>
__global__ __launch_bounds__(256) void amax_kernel(const uint16_t* __restrict__ X, size_t n8, int relu, unsigned* __restrict__ out) {
  float m = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float f[8];
    unpack8<DT>(*(const uint4*)(X + i * 8), f);
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, relu ? f[e] : fabsf(f[e]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}
hipError_t launch_amax(int mode, const void* X, size_t n, int relu, unsigned* amax_bits, hipStream_t stream) {
  if (n % 8 != 0) return hipErrorInvalidValue;
  const size_t n8 = n / 8;
  const int grid = (int)std::min<size_t>((n8 + 255) / 256, 4096);
  if (mode == MODE_FP16 || mode == MODE_FP16X3)
    hipLaunchKernelGGL(amax_kernel<DT_FP16>, dim3(grid), dim3(256), 0, stream, (const uint16_t*)X, n8, relu, amax_bits);
  else
    hipLaunchKernelGGL(amax_kernel<DT_BF16>, dim3(grid), dim3(256), 0, stream, (const uint16_t*)X, n8, relu, amax_bits);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------ range check
// Sets *flag |= 1 when a 16-bit tensor holds an Inf or a NaN (fp16 planes overflow at 65504 and nothing in the forward
// clamps).  The engine scans ONE tensor per forward -- the first head conv's output H0, where every decoder path has been
// merged through residual adds (a non-finite value upstream in the ViT blocks or the decoder reaches it; ReLUs on the way
// map NaN to 0, the residual adds next to them do not).  Integer test on the packed pairs: the exponent field is all ones
// iff magnitude + (1 << mantissa bits) carries into the sign position.
template <int DT>
__global__ __launch_bounds__(256) void nonfinite_scan_kernel(const uint4* __restrict__ X, size_t n8, unsigned* __restrict__ flag) {
  constexpr uint32_t ADD = DT == DT_FP16 ? 0x04000400u : 0x00800080u;
  uint32_t bad = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const uint4 v = X[i];
    bad |= ((v.x & 0x7fff7fffu) + ADD) | ((v.y & 0x7fff7fffu) + ADD) | ((v.z & 0x7fff7fffu) + ADD) | ((v.w & 0x7fff7fffu) + ADD);
  }
  if (bad & 0x80008000u) atomicOr(flag, 1u);
}
hipError_t launch_nonfinite_scan(int mode, const void* X, size_t n, unsigned* flag, hipStream_t stream) {
  if (n % 8 != 0 || flag == nullptr) return hipErrorInvalidValue;
  const size_t n8 = n / 8;
  const int grid = (int)std::min<size_t>((n8 + 255) / 256, 2048);
  if (mode == MODE_FP16 || mode == MODE_FP16X3)
    hipLaunchKernelGGL(nonfinite_scan_kernel<DT_FP16>, dim3(grid), dim3(256), 0, stream, (const uint4*)X, n8, flag);
  else
    hipLaunchKernelGGL(nonfinite_scan_kernel<DT_BF16>, dim3(grid), dim3(256), 0, stream, (const uint4*)X, n8, flag);
  return hipGetLastError();
}

// debug (dptx_debug_arena_checksums): *out += sum of the 32-bit words of [p, p + bytes) as a 64-bit integer
__global__ __launch_bounds__(256) void checksum_kernel(const uint32_t* __restrict__ p, size_t n32, unsigned long long* __restrict__ out) {
  unsigned long long s = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n32; i += (size_t)gridDim.x * 256) s += p[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0 && s != 0) atomicAdd(out, s);
}
// dptx_probe_stream_overlap: one wave that does nothing for `ticks` ticks of the 100 MHz wall clock (s_memrealtime) -- long
// enough to dominate launch overheads, small enough (one wave, asleep) not to disturb anything that shares the chip
__global__ __launch_bounds__(64) void spin_kernel(long long ticks) {
  const long long t0 = (long long)wall_clock64();
  while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
hipError_t launch_spin(long long ticks, hipStream_t stream) {
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, stream, ticks);
  return hipGetLastError();
}

hipError_t launch_checksum(const void* p, size_t bytes, unsigned long long* out, hipStream_t stream) {
  const size_t n32 = bytes / 4;
  if (n32 == 0) return hipSuccess;
  const int grid = (int)std::min<size_t>((n32 + 255) / 256, 1024);
  hipLaunchKernelGGL(checksum_kernel, dim3(grid), dim3(256), 0, stream, (const uint32_t*)p, n32, out);
  return hipGetLastError();
}

// ----------------------------------------------------------------- head: conv1x1 32->C (+ReLU)
// dpt_depth.py:96-98.  X [B*HW][32] 16-bit (already ReLU'd) -> y NCHW fp32 [B][C][HW].
template <int DT, int PL>
__global__ __launch_bounds__(256) void head_out_kernel(const uint16_t* __restrict__ X, const float* __restrict__ w,
                                                       const float* __restrict__ bias, void* __restrict__ y, int io, int B,
                                                       int HW, int Cout, int relu, long long plane) {
  const long long total = (long long)B * HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    float f[32];
#pragma unroll
    for (int q = 0; q < 4; ++q) load8f<DT, PL>(X + i * 32 + q * 8, plane, f + 8 * q);
    const long long b = i / HW, p = i - b * HW;
    for (int c = 0; c < Cout; ++c) {
      float acc = bias[c];
#pragma unroll
      for (int k = 0; k < 32; ++k) acc += w[c * 32 + k] * f[k];
      if (relu) acc = fmaxf(acc, 0.f);
      io_store(y, (b * Cout + c) * HW + p, acc, io);
    }
  }
}

hipError_t launch_head_out(int mode, const void* X, const float* w, const float* b, void* y, int io, int B, int HW, int Cout,
                           int relu, Planes pl, hipStream_t stream) {
  const long long total = (long long)B * HW;
  const int grid = (int)min((total + 255) / 256, (long long)16384);
  DPTX_DISPATCH_MODE(mode, hipLaunchKernelGGL((head_out_kernel<DT, PL>), dim3(grid), dim3(256), 0, stream, (const uint16_t*)X, w, b,
                                              y, io, B, HW, Cout, relu, pl.act));
  return hipGetLastError();
}

// ------------------------------------------------------------------------------- cls rows
// vit.py:141-147: x = cat(cls, patches) + pos_embed  ->  X[b*S + 0] = cls + pos[0]
// one block per image, 128 threads per 128-column block of the row (C % 128 == 0)
template <int DT>
__global__ void cls_rows_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ X, int S, int C,
                                uint16_t* __restrict__ X16, float* __restrict__ stats, uint8_t* __restrict__ X8, float q_scale) {
  __shared__ float red[2][16];
  const int b = blockIdx.x;
  for (int c0 = 0; c0 < C; c0 += blockDim.x) {  // blockDim.x is a multiple of 128
    const int c = c0 + threadIdx.x;
    float v = 0.f;
    if (c < C) {
      v = cls[c] + pos[c];
      if (X) X[(long long)b * S * C + c] = v;
      if (X16) X16[(long long)b * S * C + c] = T16<DT>::fromf(v);
      // fp8 ViT: the e4m3 copy of the token stream (block 0's qkv operand) needs its cls rows too
      if (X8) X8[(long long)b * S * C + c] = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(v * q_scale, 0.f, 0, false) & 0xff);
    }
    if (stats) {  // fixed-order reduction: wave sums, then the two waves of a 128-column block
      const float sm = wave_sum(v), sq = wave_sum(v * v);
      const int w = threadIdx.x >> 6;
      if ((threadIdx.x & 63) == 0) { red[0][w] = sm; red[1][w] = sq; }
      __syncthreads();
      if ((threadIdx.x & 127) == 0 && c < C)
        ((float2*)stats)[(long long)b * S * 8 + (c >> 7)] = make_float2(red[0][w] + red[0][w + 1], red[1][w] + red[1][w + 1]);
      __syncthreads();
    }
  }
}
hipError_t launch_cls_rows(int mode, const float* cls, const float* pos, float* X, int B, int S, int C, void* X16, float* stats,
                           hipStream_t stream, void* X8, float q_scale) {
  if (C % 128 != 0) return hipErrorInvalidValue;
  const int nt = C >= 1024 ? 1024 : (C >= 256 ? 256 : 128);
  if (mode == MODE_FP16 || mode == MODE_FP16X3)
    hipLaunchKernelGGL(cls_rows_kernel<DT_FP16>, dim3(B), dim3(nt), 0, stream, cls, pos, X, S, C, (uint16_t*)X16, stats, (uint8_t*)X8, q_scale);
  else
    hipLaunchKernelGGL(cls_rows_kernel<DT_BF16>, dim3(B), dim3(nt), 0, stream, cls, pos, X, S, C, (uint16_t*)X16, stats, (uint8_t*)X8, q_scale);
  return hipGetLastError();
}

// ------------------------------------------------------------------- pos_embed resize
// vit.py:102-116 _resize_pos_embed: the 24x24 grid part of pos_embed is resampled to (gh, gw) with
// F.interpolate(mode="bilinear", align_corners=False); the cls row is copied.  ATen's fp32 index math:
// src = max((dst + 0.5) * (in / out) - 0.5, 0); i0 = int(src); i1 = i0 + (i0 < in - 1); l1 = src - i0.
__global__ __launch_bounds__(256) void pos_resize_kernel(const float* __restrict__ src, float* __restrict__ dst, int g_old,
                                                         int gh, int gw, int C) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (gh * gw + 1) * C) return;
  const int c = i % C, t = i / C;
  if (t == 0) { dst[i] = src[c]; return; }
  const int oy = (t - 1) / gw, ox = (t - 1) - oy * gw;
  const float ry = (float)g_old / (float)gh, rx = (float)g_old / (float)gw;
  const float sy = fmaxf(((float)oy + 0.5f) * ry - 0.5f, 0.f), sx = fmaxf(((float)ox + 0.5f) * rx - 0.5f, 0.f);
  const int y0 = (int)sy, x0 = (int)sx;
  const int y1 = y0 + (y0 < g_old - 1 ? 1 : 0), x1 = x0 + (x0 < g_old - 1 ? 1 : 0);
  const float ly1 = sy - (float)y0, lx1 = sx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const float* g = src + C + c;
  dst[i] = ly0 * (lx0 * g[(y0 * g_old + x0) * C] + lx1 * g[(y0 * g_old + x1) * C]) +
           ly1 * (lx0 * g[(y1 * g_old + x0) * C] + lx1 * g[(y1 * g_old + x1) * C]);
}
hipError_t launch_pos_resize(const float* src, float* dst, int g_old, int gh, int gw, int C, hipStream_t stream) {
  hipLaunchKernelGGL(pos_resize_kernel, dim3(((gh * gw + 1) * C + 255) / 256), dim3(256), 0, stream, src, dst, g_old, gh, gw, C);
  return hipGetLastError();
}

// ----------------------------------------------------- cls half of ProjectReadout (vit.py:44-47)
// cat(tok, cls) @ W^T = tok @ W[:, :768]^T + cls @ W[:, 768:]^T : the second term is one
// vector per image, computed here and consumed as a per-image bias by the token GEMM.
// One wave per (image, output feature).
// x16 != 0: x is the 16-bit token stream (single-pass dtypes keep no fp32 copy of it)
template <int DT, int PL>
__global__ __launch_bounds__(256) void readout_cls_kernel(const float* __restrict__ x, long long x_stride,
                                                          const uint16_t* __restrict__ W, int ldw, int w_off,
                                                          const float* __restrict__ bias, float* __restrict__ out, int B, int N,
                                                          int K, long long wplane, int x16) {
  const int lane = threadIdx.x & 63;
  const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (idx >= B * N) return;
  const int b = idx / N, n = idx - b * N;
  const float* xr = x + (long long)b * x_stride;
  const uint16_t* wr = W + (long long)n * ldw + w_off;
  float acc = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    float4 xv;
    if (x16) {
      const uint2 xh = *(const uint2*)((const uint16_t*)x + (long long)b * x_stride + k);
      xv = make_float4(T16<DT>::tof((uint16_t)(xh.x & 0xffffu)), T16<DT>::tof((uint16_t)(xh.x >> 16)),
                       T16<DT>::tof((uint16_t)(xh.y & 0xffffu)), T16<DT>::tof((uint16_t)(xh.y >> 16)));
    } else {
      xv = *(const float4*)(xr + k);
    }
    const uint2 wv = *(const uint2*)(wr + k);
    float w0 = T16<DT>::tof((uint16_t)(wv.x & 0xffffu)), w1 = T16<DT>::tof((uint16_t)(wv.x >> 16));
    float w2 = T16<DT>::tof((uint16_t)(wv.y & 0xffffu)), w3 = T16<DT>::tof((uint16_t)(wv.y >> 16));
    if (PL == 2) {
      const uint2 wl = *(const uint2*)(wr + wplane + k);
      w0 += T16<DT>::tof((uint16_t)(wl.x & 0xffffu)); w1 += T16<DT>::tof((uint16_t)(wl.x >> 16));
      w2 += T16<DT>::tof((uint16_t)(wl.y & 0xffffu)); w3 += T16<DT>::tof((uint16_t)(wl.y >> 16));
    }
    acc += xv.x * w0 + xv.y * w1 + xv.z * w2 + xv.w * w3;
  }
  acc = wave_sum(acc);
  if (lane == 0) out[idx] = acc + (bias ? bias[n] : 0.f);
}

hipError_t launch_readout_cls(int mode, const float* x, long long x_stride, const void* W, int ldw, int w_off,
                              const float* bias, float* out, int B, int N, int K, Planes pl, hipStream_t stream, int x16) {
  if (K % 4 != 0 || w_off % 4 != 0 || ldw % 4 != 0) return hipErrorInvalidValue;
  dim3 grid((B * N + 3) / 4);
  DPTX_DISPATCH_MODE(mode, hipLaunchKernelGGL((readout_cls_kernel<DT, PL>), grid, dim3(256), 0, stream, x, x_stride,
                                              (const uint16_t*)W, ldw, w_off, bias, out, B, N, K, pl.w, x16));
  return hipGetLastError();
}

// ------------------------------------------------------------------- fp32 -> 16-bit (planes)
// The ProjectReadout GEMM consumes the fp32 token stream: it is rounded (or split into hi/lo
// planes) once here so that the GEMM can use the direct-to-LDS path.
template <int DT, int PL>
__global__ __launch_bounds__(256) void cast_f32_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t n8,
                                                       long long plane) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const float4 a = *(const float4*)(src + i * 8), b = *(const float4*)(src + i * 8 + 4);
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    store8f<DT, PL>(dst + i * 8, plane, f);
  }
}
hipError_t launch_cast_f32(int mode, const float* src, void* dst, size_t n, Planes pl, hipStream_t stream) {
  if (n % 8 != 0) return hipErrorInvalidValue;
  const int grid = (int)min((n / 8 + 255) / 256, (size_t)8192);
  DPTX_DISPATCH_MODE(mode, hipLaunchKernelGGL((cast_f32_kernel<DT, PL>), dim3(grid), dim3(256), 0, stream, src, (uint16_t*)dst, n / 8,
                                              pl.act));
  return hipGetLastError();
}

// ------------------------------------------------------------ 16x16 patches (DPT-Large patch embedding)
// timm PatchEmbed of vit_large_patch16_384 (vit.py:133 reaches it as patch_embed.proj): Conv2d(3, D, 16, stride 16).
// x NCHW [B,3,H,W] (fp32 / bf16 / fp16) -> P[B*(H/16)*(W/16)][768] 16-bit with k = (c, ky, kx), the order of the flattened
// conv weight, so the convolution is a dense GEMM P * W^T.  A thread owns 8 consecutive kx of one (patch, c, ky).
template <int DT, int PL>
__global__ __launch_bounds__(256) void patchify16_kernel(const void* __restrict__ x, int io, uint16_t* __restrict__ P, int B, int H,
                                                         int W, long long plane) {
  const int gh = H >> 4, gw = W >> 4;
  const long long total = (long long)B * gh * gw * 96;  // 96 8-element vectors per patch row (3 * 16 * 16 / 8)
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int v = (int)(i % 96);
    const long long patch = i / 96;
    const int px = (int)(patch % gw), py = (int)((patch / gw) % gh), b = (int)(patch / ((long long)gw * gh));
    const int c = v >> 5, ky = (v >> 1) & 15, kx0 = (v & 1) * 8;
    const long long src = (((long long)b * 3 + c) * H + py * 16 + ky) * W + px * 16 + kx0;
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = io_load(x, src + e, io);
    store8f<DT, PL>(P + patch * 768 + v * 8, plane, f);
  }
}
hipError_t launch_patchify16(int mode, const void* x, int io, void* P, int B, int H, int W, Planes pl, hipStream_t stream) {
  if (H % 16 != 0 || W % 16 != 0) return hipErrorInvalidValue;
  const long long total = (long long)B * (H / 16) * (W / 16) * 96;
  const int grid = (int)min((total + 255) / 256, (long long)16384);
  DPTX_DISPATCH_MODE(mode, hipLaunchKernelGGL((patchify16_kernel<DT, PL>), dim3(grid), dim3(256), 0, stream, x, io, (uint16_t*)P, B, H,
                                              W, pl.act));
  return hipGetLastError();
}

// --------------------------------------------------- depth to space (ConvTranspose2d with kernel == stride)
// act_postprocess1/2 of DPT-Large (vit.py:218-227, 241-250): ConvTranspose2d(C, C, k, stride k) has one tap per output
// pixel, i.e. it is a GEMM [B*h*w, C] x [C, k*k*C] followed by this re-layout:
//   G[(b, y, x)][(dy*k + dx)*C + c]  ->  Y[b][y*k + dy][x*k + dx][c]      (NHWC, 16-bit)
template <int PL>
__global__ __launch_bounds__(256) void depth_to_space_kernel(const uint16_t* __restrict__ G, uint16_t* __restrict__ Y, int B, int h,
                                                             int w, int k, int C, long long plane) {
  const int cvec = C >> 3;
  const long long total = (long long)B * h * w * k * k * cvec;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int v = (int)(i % cvec);
    long long r = i / cvec;
    const int tap = (int)(r % (k * k));
    r /= (k * k);
    const int x = (int)(r % w), y = (int)((r / w) % h), b = (int)(r / ((long long)w * h));
    const int dy = tap / k, dx = tap - dy * k;
    const long long src = (((long long)b * h + y) * w + x) * ((long long)k * k * C) + (long long)tap * C + v * 8;
    const long long dst = ((((long long)b * h * k + y * k + dy) * (w * k)) + x * k + dx) * C + v * 8;
    *(uint4*)(Y + dst) = *(const uint4*)(G + src);
    if (PL == 2) *(uint4*)(Y + plane + dst) = *(const uint4*)(G + plane + src);
  }
}
hipError_t launch_depth_to_space(int mode, const void* G, void* Y, int B, int h, int w, int k, int C, Planes pl, hipStream_t stream) {
  if (C % 8 != 0 || k < 1) return hipErrorInvalidValue;
  const long long total = (long long)B * h * w * k * k * (C / 8);
  const int grid = (int)min((total + 255) / 256, (long long)16384);
  if (mode_is_x3(mode))
    hipLaunchKernelGGL((depth_to_space_kernel<2>), dim3(grid), dim3(256), 0, stream, (const uint16_t*)G, (uint16_t*)Y, B, h, w, k, C, pl.act);
  else
    hipLaunchKernelGGL((depth_to_space_kernel<1>), dim3(grid), dim3(256), 0, stream, (const uint16_t*)G, (uint16_t*)Y, B, h, w, k, C, pl.act);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------ tap export
template <int DT, int PL>
__global__ void to_f32_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst, size_t n, long long plane) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = T16<DT>::tof(src[i]) + (PL == 2 ? T16<DT>::tof(src[i + plane]) : 0.f);
}
hipError_t launch_to_f32(int mode, const void* src, float* dst, size_t n, Planes pl, hipStream_t stream) {
  const int grid = (int)min((n + 255) / 256, (size_t)8192);
  DPTX_DISPATCH_MODE(mode, hipLaunchKernelGGL((to_f32_kernel<DT, PL>), dim3(grid), dim3(256), 0, stream, (const uint16_t*)src, dst, n,
                                              pl.act));
  return hipGetLastError();
}

}  // namespace dptx
