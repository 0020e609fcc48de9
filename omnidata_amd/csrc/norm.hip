// norm.hip -- LayerNorm (ViT blocks, eps 1e-6) and GroupNorm(32) (ResNetV2 stem/bottlenecks,
// eps 1e-5) for the DPT-Hybrid forward.  HBM-bound kernels: 16-B vector accesses, fp32
// statistics, wave-shuffle / fixed-order reductions (results are bit-reproducible run to run).
#include "common.h"
#include "kernels.h"

namespace dptx {

// ------------------------------------------------------------------------------ LayerNorm
// x fp32 [M][C] (the fp32 residual stream) -> y 16-bit [M][C].  One wave per row; C = 768:
// each lane owns 3 float4.  Two-pass (mean, then centred variance) in registers.
template <int DT, int PL, int VPL>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, uint16_t* __restrict__ y,
                                                        int M, int C, float eps, long long plane) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (long long)row * C;
  float4 v[VPL];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    v[j] = *(const float4*)(xr + (j * 64 + lane) * 4);
    s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
  }
  const float mean = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
    q += (a * a + b * b) + (c * c + d * d);
  }
  const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
  uint16_t* yr = y + (long long)row * C;
#pragma unroll
  for (int j = 0; j < VPL; ++j) {
    const int c0 = (j * 64 + lane) * 4;
    const float4 g = *(const float4*)(gamma + c0), bb = *(const float4*)(beta + c0);
    const float o0 = (v[j].x - mean) * rstd * g.x + bb.x, o1 = (v[j].y - mean) * rstd * g.y + bb.y;
    const float o2 = (v[j].z - mean) * rstd * g.z + bb.z, o3 = (v[j].w - mean) * rstd * g.w + bb.w;
    uint2 w;
    w.x = T16<DT>::pack2(o0, o1);
    w.y = T16<DT>::pack2(o2, o3);
    *(uint2*)(yr + c0) = w;
    if (PL == 2) {
      uint2 l;
      l.x = T16<DT>::pack2(o0 - T16<DT>::tof((uint16_t)(w.x & 0xffffu)), o1 - T16<DT>::tof((uint16_t)(w.x >> 16)));
      l.y = T16<DT>::pack2(o2 - T16<DT>::tof((uint16_t)(w.y & 0xffffu)), o3 - T16<DT>::tof((uint16_t)(w.y >> 16)));
      *(uint2*)(yr + plane + c0) = l;
    }
  }
}

hipError_t launch_layernorm(int mode, const float* x, const float* gamma, const float* beta, void* y, int M, int C,
                            float eps, Planes pl, hipStream_t stream) {
  if (C != 768 && C != 1024) return hipErrorInvalidValue;  // ViT-B (hybrid) / ViT-L widths: 3 or 4 float4 per lane
  dim3 grid((M + 3) / 4);
  if (C == 768) {
    DPTX_DISPATCH_MODE(mode, hipLaunchKernelGGL((layernorm_kernel<DT, PL, 3>), grid, dim3(256), 0, stream, x, gamma, beta,
                                                (uint16_t*)y, M, C, eps, pl.act));
  } else {
    DPTX_DISPATCH_MODE(mode, hipLaunchKernelGGL((layernorm_kernel<DT, PL, 4>), grid, dim3(256), 0, stream, x, gamma, beta,
                                                (uint16_t*)y, M, C, eps, pl.act));
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------ GroupNorm
// X NHWC 16-bit [B][HW][C], 32 groups of cpg = C/32 adjacent channels.
// Stats pass: grid (chunks, B); a block reduces `pix` pixels x C channels to 32 x (sum, sumsq)
// in a fixed order and writes partial[b][chunk][g][2].  The apply pass re-reduces the chunk
// partials in double (fixed order) in its prologue, so no atomics and no finalize launch.
constexpr int GN_G = 32;

// pixels per block: ~32 KB of the tensor (256 px at C = 64 ... 16 px at C = 1024) -- a function of the layer only, never
// of the batch, so that the partial-sum order (and with it every output bit) is the same at any batch size
static inline int gn_pix(int C) {
  const int pix = 16384 / C;
  return pix < 16 ? 16 : (pix > 256 ? 256 : pix);
}
int gn_chunks(int HW, int C) {
  const int pix = gn_pix(C);
  return (HW + pix - 1) / pix;
}

template <int DT, int PL>
__global__ __launch_bounds__(256) void gn_stats_kernel(const uint16_t* __restrict__ X, float* __restrict__ partial,
                                                       int HW, int C, int pix, long long plane) {
  __shared__ float red[256 * 8];  // per thread: up to 4 (sum, sumsq) pairs
  const int tid = threadIdx.x;
  const int cvec = C >> 3;            // 16-B vectors per pixel (8..128)
  const int cpg = C >> 5;             // channels per group (2..32)
  const int spv = cpg >= 8 ? 1 : 8 / cpg;  // groups (slots) per vector: 1, 2 or 4
  const int v = tid % cvec, p0 = tid / cvec, pstep = 256 / cvec;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int pbeg = chunk * pix;
  const int pend = min(pbeg + pix, HW);
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
  const uint16_t* base = X + ((long long)b * HW) * C + v * 8;
  for (int p = pbeg + p0; p < pend; p += pstep) {
    float f[8];
    load8f<DT, PL>(base + (long long)p * C, plane, f);
    if (spv == 1) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { s[0] += f[e]; q[0] += f[e] * f[e]; }
    } else if (spv == 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[0] += f[e]; q[0] += f[e] * f[e]; s[1] += f[4 + e]; q[1] += f[4 + e] * f[4 + e]; }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[e] += f[2 * e] + f[2 * e + 1]; q[e] += f[2 * e] * f[2 * e] + f[2 * e + 1] * f[2 * e + 1]; }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[tid * 8 + 2 * e] = s[e]; red[tid * 8 + 2 * e + 1] = q[e]; }
  __syncthreads();
  if (tid < GN_G) {
    // group g <- vectors [g*cpg/8, (g+1)*cpg/8) slot 0 (cpg >= 8) or vector g/spv slot g%spv
    const int g = tid;
    int v0, v1, slot;
    if (spv == 1) { v0 = g * (cpg >> 3); v1 = v0 + (cpg >> 3); slot = 0; }
    else { v0 = g / spv; v1 = v0 + 1; slot = g % spv; }
    float ss = 0.f, qq = 0.f;
    for (int pp = 0; pp < pstep; ++pp)
      for (int vv = v0; vv < v1; ++vv) {
        const int t = pp * cvec + vv;
        ss += red[t * 8 + 2 * slot];
        qq += red[t * 8 + 2 * slot + 1];
      }
    float* out = partial + (((long long)b * gridDim.x + chunk) * GN_G + g) * 2;
    out[0] = ss;
    out[1] = qq;
  }
}

hipError_t launch_gn_stats(int mode, const void* X, float* partial, int B, int HW, int C, Planes pl, hipStream_t stream) {
  if (C % 64 != 0 || C > 1024 || (256 % (C / 8)) != 0) return hipErrorInvalidValue;
  dim3 grid(gn_chunks(HW, C), B);
  DPTX_DISPATCH_MODE(mode, hipLaunchKernelGGL((gn_stats_kernel<DT, PL>), grid, dim3(256), 0, stream, (const uint16_t*)X, partial,
                                              HW, C, gn_pix(C), pl.act));
  return hipGetLastError();
}

// per-channel affine (a, d) with y = x*a + d from chunk partials; called by every thread of a 256-thread block.
// The chunk partials are summed in double by 8 slices of 32 threads (slice s takes chunks s, s+8, ...: the loads of
// one slice are independent, so the latency is nchunks/8 dependent adds instead of nchunks dependent loads) and the
// slices are combined in a fixed order: the result depends on HW and C only, never on the batch or the run.
__device__ __forceinline__ void gn_affine_to_lds(const float* __restrict__ partial, int nchunks, int b,
                                                 const float* __restrict__ gamma, const float* __restrict__ beta, int C,
                                                 int HW, float eps, float* sa, float* sd, float* smr) {
  __shared__ double sred[8][GN_G][2];
  const int tid = threadIdx.x;
  const int cpg = C >> 5;
  {
    const int g = tid & (GN_G - 1), slice = tid >> 5;
    double ss = 0.0, qq = 0.0;
    const float2* pp = (const float2*)partial + (long long)b * nchunks * GN_G + g;
    for (int c = slice; c < nchunks; c += 8) {
      const float2 v = pp[(long long)c * GN_G];
      ss += (double)v.x;
      qq += (double)v.y;
    }
    sred[slice][g][0] = ss;
    sred[slice][g][1] = qq;
  }
  __syncthreads();
  if (tid < GN_G) {
    double ss = 0.0, qq = 0.0;
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) { ss += sred[sl][tid][0]; qq += sred[sl][tid][1]; }
    const double n = (double)HW * (double)cpg;
    const double mean = ss / n;
    double var = qq / n - mean * mean;
    if (var < 0.0) var = 0.0;
    smr[2 * tid] = (float)mean;
    smr[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    const int g = c / cpg;
    const float a = smr[2 * g + 1] * gamma[c];
    sa[c] = a;
    sd[c] = beta[c] - smr[2 * g] * a;
  }
  __syncthreads();
}

template <int DT, int PL>
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnParams p, int pix, int nrec, long long plane) {
  __shared__ float sa[1024], sd[1024], ra[1024], rd[1024], smr[64];
  const int b = blockIdx.y;
  gn_affine_to_lds(p.partial, nrec, b, p.gamma, p.beta, p.C, p.HW, p.eps, sa, sd, smr);
  const bool r_gn = (p.R != nullptr) && (p.r_gamma != nullptr);
  if (r_gn) gn_affine_to_lds(p.r_partial, nrec, b, p.r_gamma, p.r_beta, p.C, p.HW, p.eps, ra, rd, smr);
  const int cvec = p.C >> 3;
  const int pbeg = blockIdx.x * pix;
  const int pend = min(pbeg + pix, p.HW);
  const long long img = (long long)b * p.HW * p.C;
  const uint16_t* X = (const uint16_t*)p.X + img;
  const uint16_t* R = p.R ? (const uint16_t*)p.R + img : nullptr;
  uint16_t* Y = (uint16_t*)p.Y + img;
  const int total = (pend - pbeg) * cvec;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int pi = i / cvec, v = i - pi * cvec;
    const long long off = (long long)(pbeg + pi) * p.C + v * 8;
    float f[8];
    load8f<DT, PL>(X + off, plane, f);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = f[e] * sa[v * 8 + e] + sd[v * 8 + e];
    if (R != nullptr) {
      float r[8];
      load8f<DT, PL>(R + off, plane, r);
      if (r_gn) {
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = r[e] * ra[v * 8 + e] + rd[v * 8 + e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] += r[e];
    }
    if (p.relu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
    }
    store8f<DT, PL>(Y + off, plane, f);
  }
}

hipError_t launch_gn_apply(int mode, const GnParams& p, Planes pl, hipStream_t stream) {
  if (p.C % 64 != 0 || p.C > 1024) return hipErrorInvalidValue;
  const int nch = gn_chunks(p.HW, p.C);
  const int nrec = p.nrec > 0 ? p.nrec : nch;
  // every block re-reduces its image's partial records in its prologue (nrec x 256 B from L2): with the fine records of
  // the GEMM epilogue (one per 32 rows) that costs more than the block's own share of the map unless blocks are fat --
  // aim at ~1024 blocks per launch, at least 4 per image.  The partition is free: the apply is elementwise.
  int nb = nch;
  if (p.nrec > 0) {
    int want = 1024 / (p.B > 0 ? p.B : 1);
    want = want < 4 ? 4 : want;
    nb = nch < want ? nch : want;
  }
  const int pix = (p.HW + nb - 1) / nb;
  dim3 grid((p.HW + pix - 1) / pix, p.B);
  DPTX_DISPATCH_MODE(mode, hipLaunchKernelGGL((gn_apply_kernel<DT, PL>), grid, dim3(256), 0, stream, p, pix, nrec, pl.act));
  return hipGetLastError();
}

// stem: GN + ReLU + MaxPool2dSame(3, stride 2): pad (0,1) bottom/right; padded taps are skipped
// (equivalent to timm's -inf padding).  One thread = one output pixel x 8 channels.
template <int DT, int PL>
__global__ __launch_bounds__(256) void gn_relu_maxpool_kernel(const uint16_t* __restrict__ X, uint16_t* __restrict__ Y,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ partial, int nchunks, int H, int W,
                                                              int C, float eps, long long plane) {
  __shared__ float sa[1024], sd[1024], smr[64];
  const int b = blockIdx.y;
  gn_affine_to_lds(partial, nchunks, b, gamma, beta, C, H * W, eps, sa, sd, smr);
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int cvec = C >> 3;
  const int total = Ho * Wo * cvec;
  const uint16_t* Xi = X + (long long)b * H * W * C;
  uint16_t* Yo = Y + (long long)b * Ho * Wo * C;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int v = i % cvec, po = i / cvec;
    const int oy = po / Wo, ox = po - oy * Wo;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = 0.f;  // post-ReLU values are >= 0
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = 2 * oy + dy;
      if (iy >= H) continue;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int ix = 2 * ox + dx;
        if (ix >= W) continue;
        float f[8];
        load8f<DT, PL>(Xi + ((long long)iy * W + ix) * C + v * 8, plane, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], f[e] * sa[v * 8 + e] + sd[v * 8 + e]);
      }
    }
    store8f<DT, PL>(Yo + (long long)po * C + v * 8, plane, m);
  }
}

hipError_t launch_gn_relu_maxpool(int mode, const void* X, void* Y, const float* gamma, const float* beta,
                                  const float* partial, int B, int H, int W, int C, float eps, Planes pl, hipStream_t stream) {
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const int total = Ho * Wo * (C / 8);
  // few, fat blocks: every block re-reduces the chunk partials of its image in its prologue
  dim3 grid(min((total + 255) / 256, 64), B);
  const int nch = gn_chunks(H * W, C);
  DPTX_DISPATCH_MODE(mode, hipLaunchKernelGGL((gn_relu_maxpool_kernel<DT, PL>), grid, dim3(256), 0, stream, (const uint16_t*)X,
                                              (uint16_t*)Y, gamma, beta, partial, nch, H, W, C, eps, pl.act));
  return hipGetLastError();
}

}  // namespace dptx
