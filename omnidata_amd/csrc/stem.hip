// stem.hip -- fused 7x7 stride-2 TF-SAME stem convolution of the ResNetV2 hybrid backbone
// (timm StdConv2dSame 3->64; vit.py:128-131 reaches it through patch_embed.backbone).
//
// x NCHW fp32 (or bf16 / fp16: DPTX_IO_*) [B,3,H,W]  ->  y NHWC 16-bit [B,H/2,W/2,64]   (raw conv output; GroupNorm+ReLU+MaxPool follow)
//
// No im2col buffer: a block owns 4 output rows x 64 output columns.  Its 13 x 133 x 3 input patch is converted to
// 16-bit once and kept in LDS (10.6 KB); the implicit-GEMM K axis is ordered (c, ky, kx) with kx padded 7 -> 8, so
// that one MFMA operand chunk (8 consecutive k) is 8 CONSECUTIVE input columns of one (channel, row) -- 16
// contiguous, 4-byte-aligned bytes of the patch, fetched with four conflict-free ds_read_b32 (neighbouring output
// pixels are 4 bytes apart).  K = 3*7*8 = 168, padded to 176 = 11 MFMA k-steps; the weight matrix [64][176]
// (22.5 KB, zero in the padded taps) is read straight from L1/L2.  v_mfma_f32_32x32x16: wave w computes output row
// w of the block (64 pixels x 64 channels).  Replaces im2col (453 MB written + read per 32 images) + a K=192 GEMM.
#include "common.h"
#include "kernels.h"

namespace dptx {

constexpr int STEM_KP = 176;       // packed K of the stem weight rows
constexpr int ST_ROWS = 13, ST_PITCH = 136;  // patch rows, patch row pitch (elements)
constexpr int ST_PATCH = 3 * ST_ROWS * ST_PITCH;  // elements per plane

template <int DT, int PL>
__global__ __launch_bounds__(256, 2) void stem_conv_kernel(const void* __restrict__ x, int io, const uint16_t* __restrict__ Wt,
                                                           uint16_t* __restrict__ y, int H, int W, int pad_t, int pad_l,
                                                           long long act_plane, long long w_plane) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint16_t* patch = (uint16_t*)smem;  // [PL][3][13][136]
  const int Ho = H >> 1, Wo = W >> 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int tiles_x = (Wo + 63) / 64;   // the last column tile may be partial (W any multiple of 8)
  const int ox0 = (blockIdx.x % tiles_x) * 64, oy0 = (blockIdx.x / tiles_x) * 4, b = blockIdx.y;
  const int iy0 = 2 * oy0 - pad_t, ix0 = 2 * ox0 - pad_l;

  // ---- input patch -> LDS (zero outside the image and in the pitch padding).  A wave owns one (channel, row) line of the
  // patch per pass and moves it as column PAIRS: the patch starts at an even column (ix0 = 128 k - pad_l, pad_l = 2) and W is
  // even, so a pair is either inside the image or outside it -- one 8-byte (fp32) / 4-byte (16-bit) load and one ds_write_b32
  // per pair instead of two scalar loads, two conversions and four integer divisions per element
  const long long xb = (long long)b * 3 * H * W;  // element offset of image b (x is fp32, bf16 or fp16: `io`)
  for (int rc = wave; rc < 3 * ST_ROWS; rc += 4) {
    const int c = rc / ST_ROWS, r = rc - c * ST_ROWS;
    const int iy = iy0 + r;
    const bool rowok = (unsigned)iy < (unsigned)H;
    const long long rowoff = xb + ((long long)c * H + iy) * W;
    for (int cp = lane; cp < ST_PITCH / 2; cp += 64) {
      const int col = 2 * cp, ix = ix0 + col;
      float v0 = 0.f, v1 = 0.f;
      if (rowok && col < 134 && (unsigned)ix < (unsigned)W) {
        if (io == IO_FP32) {
          const float2 t = *(const float2*)((const float*)x + rowoff + ix);
          v0 = t.x; v1 = t.y;
        } else {
          const uint32_t t = *(const uint32_t*)((const uint16_t*)x + rowoff + ix);
          v0 = io == IO_BF16 ? T16<DT_BF16>::tof((uint16_t)(t & 0xffffu)) : T16<DT_FP16>::tof((uint16_t)(t & 0xffffu));
          v1 = io == IO_BF16 ? T16<DT_BF16>::tof((uint16_t)(t >> 16)) : T16<DT_FP16>::tof((uint16_t)(t >> 16));
        }
      }
      const uint32_t hi = T16<DT>::pack2(v0, v1);
      *(uint32_t*)(patch + rc * ST_PITCH + col) = hi;
      if (PL == 2)
        *(uint32_t*)(patch + ST_PATCH + rc * ST_PITCH + col) =
            T16<DT>::pack2(v0 - T16<DT>::tof((uint16_t)(hi & 0xffffu)), v1 - T16<DT>::tof((uint16_t)(hi >> 16)));
    }
  }
  __syncthreads();

  f32x16_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int ks = 0; ks < STEM_KP / 16; ++ks) {
    const int q = 2 * ks + lh;                 // k chunk 0..21; chunk 21 is all-zero weights
    const int qq = q < 21 ? q : 0;
    const int c = qq / 7, ky = qq - c * 7;
    u32x4_t af[2], al[2], bf[2], bl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int cx = i * 32 + lr;              // output column inside the block
      const uint32_t* src = (const uint32_t*)(patch + ((c * ST_ROWS + 2 * wave + ky) * ST_PITCH + 2 * cx));
      af[i] = u32x4_t{src[0], src[1], src[2], src[3]};
      if (PL == 2) {
        const uint32_t* sl = src + ST_PATCH / 2;
        al[i] = u32x4_t{sl[0], sl[1], sl[2], sl[3]};
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint16_t* wp = Wt + (long long)(j * 32 + lr) * STEM_KP + q * 8;
      bf[j] = *(const u32x4_t*)wp;
      if (PL == 2) bl[j] = *(const u32x4_t*)(wp + w_plane);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (PL == 2) {
          acc[i][j] = T16<DT>::mfma32(al[i], bf[j], acc[i][j]);
          acc[i][j] = T16<DT>::mfma32(af[i], bl[j], acc[i][j]);
        }
        acc[i][j] = T16<DT>::mfma32(af[i], bf[j], acc[i][j]);
      }
  }

  // ---- epilogue: accumulators -> LDS [256 pixels][64 ch] fp32 -> coalesced NHWC rows (128 B per pixel)
  __syncthreads();
  float* ct = (float*)smem;
  constexpr int CT_PITCH = 68;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int pl_ = wave * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;  // pixel = (row wave, column)
        ct[pl_ * CT_PITCH + j * 32 + lr] = acc[i][j][r];
      }
  __syncthreads();
  const int cn = tid & 7;
#pragma unroll 2
  for (int p = tid >> 3; p < 256; p += 32) {
    const int r = p >> 6, c = p & 63;
    float v[8];
    const float4 a0 = *(const float4*)(ct + p * CT_PITCH + cn * 8), a1 = *(const float4*)(ct + p * CT_PITCH + cn * 8 + 4);
    v[0] = a0.x; v[1] = a0.y; v[2] = a0.z; v[3] = a0.w; v[4] = a1.x; v[5] = a1.y; v[6] = a1.z; v[7] = a1.w;
    if (ox0 + c < Wo) store8f<DT, PL>(y + (((long long)b * Ho + oy0 + r) * Wo + ox0 + c) * 64 + cn * 8, act_plane, v);
  }
}

hipError_t launch_stem_conv(int mode, const void* x, int io, const void* Wt, void* y, int B, int H, int W, Planes pl,
                            hipStream_t stream) {
  const int Ho = H / 2, Wo = W / 2;
  if (H % 8 != 0 || W % 8 != 0) return hipErrorInvalidValue;
  const int pt = max((Ho - 1) * 2 + 7 - H, 0) / 2, plft = max((Wo - 1) * 2 + 7 - W, 0) / 2;
  dim3 grid(((Wo + 63) / 64) * (Ho / 4), B);
  const size_t smem = 256 * 68 * 4;  // C tile (69.6 KB) aliases the patch
  DPTX_DISPATCH_MODE(mode, {
    auto k = stem_conv_kernel<DT, PL>;
    ensure_dyn_smem((const void*)k, smem);
    hipLaunchKernelGGL(k, grid, dim3(256), smem, stream, x, io, (const uint16_t*)Wt, (uint16_t*)y, H, W, pt, plft, pl.act, pl.w);
  });
  return hipGetLastError();
}

}  // namespace dptx
