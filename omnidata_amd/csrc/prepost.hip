// prepost.hip -- GPU-side pre/post-processing of the reference's demo.py, so the host
// is out of the per-image loop.
//
//  pre : uint8 HWC image -> Resize(384, BILINEAR, shorter side) -> CenterCrop(384) -> ToTensor (/255)
//        [-> Normalize(0.5, 0.5) for depth]                      demo.py:74-76, 92-95, 130-138
//        The reference resizes a PIL image, i.e. Pillow's antialiased two-pass 8-bit resampler
//        (libImaging/Resample.c): per output coordinate a window [xmin, xmin+n) of triangle-filter weights,
//        normalised, converted to 22-bit fixed point, horizontal pass -> uint8 (rounded, clipped), vertical
//        pass -> uint8.  That arithmetic is restated exactly (the coefficient tables are built on the host in
//        double, like Pillow does); results are bit-identical to PIL (tests/test_prepost.py, tests/test_gpu_prepost.py).
//  post: normal: clamp(0,1)*255 -> uint8 HWC (ToPILImage truncation)      demo.py:140,150
//        depth : bicubic 384->512 (ATen, A=-0.75, align_corners=False), clamp(0,1), 1-x   demo.py:143-145
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/dptx.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;  // Pillow: 22

// Pillow precompute_coeffs + normalize_coeffs_8bpc for the bilinear (triangle, support 1) filter, box = whole image
void resample_coeffs(int in_size, int out_size, std::vector<int>& bounds, std::vector<int>& kk, int& ksize) {
  const double scale = (double)in_size / out_size;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  ksize = (int)std::ceil(support) * 2 + 1;
  bounds.assign((size_t)out_size * 2, 0);
  kk.assign((size_t)out_size * ksize, 0);
  std::vector<double> k(ksize);
  for (int xx = 0; xx < out_size; ++xx) {
    const double center = (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
      double t = (x + xmin - center + 0.5) * ss;
      if (t < 0.0) t = -t;
      const double w = t < 1.0 ? 1.0 - t : 0.0;
      k[x] = w;
      ww += w;
    }
    for (int x = 0; x < xmax; ++x) {
      if (ww != 0.0) k[x] /= ww;
      kk[(size_t)xx * ksize + x] = k[x] < 0 ? (int)(-0.5 + k[x] * (1 << PRECISION_BITS)) : (int)(0.5 + k[x] * (1 << PRECISION_BITS));
    }
    bounds[2 * xx] = xmin;
    bounds[2 * xx + 1] = xmax;
  }
}

__device__ __forceinline__ int clip8(int v) {
  v >>= PRECISION_BITS;
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// one thread = one output pixel (all channels); tables: [384][2] bounds + [384][ksize] weights per axis, already
// offset by the centre crop
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t* __restrict__ img, int H, int W, int C, int row_stride,
                                                         const int* __restrict__ bh, const int* __restrict__ kh, int ksh,
                                                         const int* __restrict__ bv, const int* __restrict__ kv, int ksv,
                                                         int depth_norm, float* __restrict__ out, int S) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= S * S) return;
  const int oy = idx / S, ox = idx - oy * S;
  const int xmin = bh[2 * ox], nx = bh[2 * ox + 1];
  const int ymin = bv[2 * oy], ny = bv[2 * oy + 1];
  const int* wh = kh + ox * ksh;
  const int* wv = kv + oy * ksv;
  int acc[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
  const int nc = C >= 3 ? 3 : 1;
  for (int yy = 0; yy < ny; ++yy) {
    const uint8_t* row = img + (size_t)(ymin + yy) * row_stride + (size_t)xmin * C;
    int h[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
    for (int xx = 0; xx < nx; ++xx) {
      const int w = wh[xx];
      for (int c = 0; c < nc; ++c) h[c] += (int)row[xx * C + c] * w;
    }
    const int wy = wv[yy];
    for (int c = 0; c < nc; ++c) acc[c] += clip8(h[c]) * wy;  // horizontal pass result is a rounded uint8
  }
  for (int c = 0; c < 3; ++c) {
    const int v8 = clip8(acc[nc == 3 ? c : 0]);  // 1-channel input is repeated (demo.py:137-138)
    float f = (float)v8 / 255.0f;                // ToTensor
    if (depth_norm) f = (f - 0.5f) / 0.5f;       // Normalize(mean=0.5, std=0.5)
    out[((size_t)c * S + oy) * S + ox] = f;
  }
}

__global__ __launch_bounds__(256) void post_normal_kernel(const float* __restrict__ y, uint8_t* __restrict__ out, int HW) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HW) return;
  for (int c = 0; c < 3; ++c) {
    float v = y[(size_t)c * HW + i];
    v = fminf(fmaxf(v, 0.f), 1.f);
    out[(size_t)i * 3 + c] = (uint8_t)(v * 255.0f);  // ToPILImage: mul(255).byte() truncates
  }
}

// ATen upsample_bicubic2d (align_corners=False): src = scale*(dst+0.5)-0.5, A = -0.75, border-clamped taps
__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
__global__ __launch_bounds__(256) void post_depth_kernel(const float* __restrict__ y, float* __restrict__ out, int S, int T) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= T * T) return;
  const int oy = idx / T, ox = idx - oy * T;
  const float scale = (float)S / (float)T;
  const float A = -0.75f;
  const float sy = scale * ((float)oy + 0.5f) - 0.5f, sx = scale * ((float)ox + 0.5f) - 0.5f;
  const int iy = (int)floorf(sy), ix = (int)floorf(sx);
  const float ty = sy - (float)iy, tx = sx - (float)ix;
  const float cy[4] = {cubic2(ty + 1.f, A), cubic1(ty, A), cubic1(1.f - ty, A), cubic2(2.f - ty, A)};
  const float cx[4] = {cubic2(tx + 1.f, A), cubic1(tx, A), cubic1(1.f - tx, A), cubic2(2.f - tx, A)};
  float acc = 0.f;
  for (int j = 0; j < 4; ++j) {
    const int yy = min(max(iy - 1 + j, 0), S - 1);
    float r = 0.f;
    for (int i = 0; i < 4; ++i) {
      const int xx = min(max(ix - 1 + i, 0), S - 1);
      r += y[(size_t)yy * S + xx] * cx[i];
    }
    acc += r * cy[j];
  }
  acc = fminf(fmaxf(acc, 0.f), 1.f);
  out[idx] = 1.f - acc;
}

struct Tables {
  int* d = nullptr;  // [bh(2S) | kh(S*ksh) | bv(2S) | kv(S*ksv)]
  int ksh = 0, ksv = 0;
};
std::mutex g_mu;
std::map<std::pair<long long, int>, Tables> g_tables;  // key: ((H<<32)|W, device)

// resized size (torchvision Resize(int)): shorter side -> S, other = int(S*long/short); then centre crop offsets
void resized_geometry(int H, int W, int S, int& oh, int& ow, int& top, int& left) {
  if ((W <= H && W == S) || (H <= W && H == S)) { oh = H; ow = W; }
  else if (W < H) { ow = S; oh = (int)((long long)S * H / W); }
  else { oh = S; ow = (int)((long long)S * W / H); }
  // torchvision CenterCrop: int(round(x / 2.0)) with Python's round-half-to-even
  top = (int)std::nearbyint((oh - S) / 2.0);
  left = (int)std::nearbyint((ow - S) / 2.0);
}

}  // namespace

extern "C" {

int dptx_resample_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* kk, int32_t kk_capacity, int32_t* ksize) {
  if (in_size < 1 || out_size < 1 || !bounds || !kk || !ksize) return DPTX_E_INVALID;
  std::vector<int> b, k;
  int ks = 0;
  resample_coeffs(in_size, out_size, b, k, ks);
  *ksize = ks;
  if ((long long)kk_capacity < (long long)out_size * ks) return DPTX_E_INVALID;
  memcpy(bounds, b.data(), b.size() * sizeof(int));
  memcpy(kk, k.data(), k.size() * sizeof(int));
  return DPTX_OK;
}

int dptx_preprocess_u8(const void* img_dev, int32_t H, int32_t W, int32_t C, int32_t row_stride_bytes, int32_t depth_normalize,
                       void* x_dev, void* stream) {
  constexpr int S = 384;
  if (!img_dev || !x_dev || H < 1 || W < 1 || (C != 1 && C != 3) || row_stride_bytes < W * C) return DPTX_E_INVALID;
  int oh, ow, top, left;
  resized_geometry(H, W, S, oh, ow, top, left);
  if (oh < S || ow < S) return DPTX_E_INVALID;  // CenterCrop would pad: not needed after Resize(384)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return DPTX_E_HIP;
  Tables t;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    const auto key = std::make_pair(((long long)H << 32) | (unsigned)W, dev);
    auto it = g_tables.find(key);
    if (it == g_tables.end()) {
      std::vector<int> bh, kh, bv, kv;
      int ksh, ksv;
      resample_coeffs(W, ow, bh, kh, ksh);
      resample_coeffs(H, oh, bv, kv, ksv);
      std::vector<int> host;
      host.insert(host.end(), bh.begin() + 2 * left, bh.begin() + 2 * (left + S));
      host.insert(host.end(), kh.begin() + (size_t)left * ksh, kh.begin() + (size_t)(left + S) * ksh);
      host.insert(host.end(), bv.begin() + 2 * top, bv.begin() + 2 * (top + S));
      host.insert(host.end(), kv.begin() + (size_t)top * ksv, kv.begin() + (size_t)(top + S) * ksv);
      Tables nt;
      nt.ksh = ksh;
      nt.ksv = ksv;
      if (hipMalloc((void**)&nt.d, host.size() * sizeof(int)) != hipSuccess) return DPTX_E_ALLOC;
      if (hipMemcpy(nt.d, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return DPTX_E_HIP;
      it = g_tables.emplace(key, nt).first;
    }
    t = it->second;
  }
  const int* bh = t.d;
  const int* kh = bh + 2 * S;
  const int* bv = kh + (size_t)S * t.ksh;
  const int* kv = bv + 2 * S;
  hipLaunchKernelGGL(preprocess_kernel, dim3((S * S + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)img_dev, H, W, C,
                     row_stride_bytes, bh, kh, t.ksh, bv, kv, t.ksv, depth_normalize, (float*)x_dev, S);
  return hipGetLastError() == hipSuccess ? DPTX_OK : DPTX_E_HIP;
}

int dptx_postprocess_normal_u8(const void* y_dev, void* rgb_u8_dev, void* stream) {
  if (!y_dev || !rgb_u8_dev) return DPTX_E_INVALID;
  constexpr int HW = 384 * 384;
  hipLaunchKernelGGL(post_normal_kernel, dim3((HW + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)y_dev,
                     (uint8_t*)rgb_u8_dev, HW);
  return hipGetLastError() == hipSuccess ? DPTX_OK : DPTX_E_HIP;
}

int dptx_postprocess_depth(const void* y_dev, void* out512_dev, void* stream) {
  if (!y_dev || !out512_dev) return DPTX_E_INVALID;
  hipLaunchKernelGGL(post_depth_kernel, dim3((512 * 512 + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)y_dev,
                     (float*)out512_dev, 384, 512);
  return hipGetLastError() == hipSuccess ? DPTX_OK : DPTX_E_HIP;
}

}  // extern "C"
