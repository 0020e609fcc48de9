// attention.hip -- fused softmax(Q K^T / sqrt(64)) V for the 12 ViT blocks (timm Attention,
// call site vit.py:150-151; N = 577 tokens, 12 heads x 64).  Scores never reach HBM.
//
// Input  qkv[B*S][3*H*64] 16-bit, feature index = which*H*64 + head*64 + dim (timm's
//        reshape(B,N,3,heads,64) packing); output out[B*S][H*64] 16-bit.
// Grid   1-D, ceil(S/128) * B * H blocks (q-block slowest: K/V sharers stay on one XCD); 4 waves x 32 query rows.
// Per 64-key tile (shared by the 4 waves through LDS):
//   K  tile [64 keys][64 d]   row-major, 16-B chunks XOR-swizzled (same scheme as gemm.hip)
//   V^T tile [64 d][64 keys]  transposed while staging (two adjacent keys per ds_write_b32),
//                             same XOR swizzle keyed on the d row, key order permuted (bits 2<->3
//                             within each 32-key half) so that the P^T accumulator registers of
//                             the first MFMA are directly the B operand of the second one -- no
//                             cross-lane movement of P.
//   S^T[key][q] = mfma32x32x16(K, Q)   (swapped operands: a lane owns ONE query column, so the
//                                       row max / row sum are in-lane + one lane^32 exchange)
//   online softmax in fp32 (exp2 with the 1/8 scale folded into the exponent constant)
//   O^T[d][q] += mfma32x32x16(V^T, P^T)
#include "common.h"
#include "kernels.h"

namespace dptx {

constexpr int ATT_D = 64;
constexpr int ATT_KT = 64;                 // keys per tile
constexpr int ATT_K_BYTES = ATT_KT * 128;  // 8 KB
constexpr int ATT_V_BYTES = ATT_D * 128;   // 8 KB
constexpr int ATT_STAGE = ATT_K_BYTES + ATT_V_BYTES;

__device__ __forceinline__ int vt_pos(int key) {  // swap bits 2 and 3
  return (key & ~12) | ((key & 4) << 1) | ((key & 8) >> 1);
}

// PL == 2 (bf16x3 / fp16x3 modes): q/k/v/p are hi+lo plane pairs and every product is 3 MFMAs
// (lo*hi + hi*lo + hi*hi); `plane` is the element distance between the planes of qkv / out.
//
// Round 3 (profiles/r02_pmc_sq.txt: MFMA busy 10 % of the wave cycles, 34 % issue stalls behind dependent MFMAs, 29 %
// waits): (a) key 0 -- the cls token -- is handled once per block on the VALU (one 64-long dot product per query, the
// online-softmax state starts at m = s0, l = 1, O = v0), so the key tiles cover keys 1..S-1: 576 = 9 x 64 of them at
// 384x384 instead of ten tiles the last of which held ONE key; (b) the two 32-key halves of a tile go through QK^T
// together (two independent MFMA chains instead of one dependent chain of four) and share ONE running-max update, one
// alpha and at most one rescale of O per tile; (c) waves whose 32 queries all lie beyond S only stage K / V.
template <int DT, int PL>
__global__ __launch_bounds__(256, 2) void attention_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                                           int S, int H, int BH, long long plane) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 stages x PL x (K tile + V^T tile)
  constexpr int STAGE = ATT_STAGE * PL;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  // 1-D grid, id = qblock * (B*H) + (b*H + head): the q-blocks that re-read one (batch, head)'s K/V have ids that
  // differ by B*H (a multiple of 8 for even B), i.e. they run on the same XCD and share its L2
  const int bh = (int)blockIdx.x % BH, qblk = (int)blockIdx.x / BH;
  const int head = bh % H, b = bh / H;
  const int ld = 3 * H * ATT_D;
  const long long row0 = (long long)b * S;
  const uint16_t* qbase = qkv + head * ATT_D;
  const uint16_t* kbase = qkv + H * ATT_D + head * ATT_D;
  const uint16_t* vbase = qkv + 2 * H * ATT_D + head * ATT_D;

  // Q fragments (B operand: lane = query column, 8 consecutive d per k-step)
  const int q = qblk * 128 + wave * 32 + lr;
  const int qc = q < S ? q : S - 1;
  const bool wave_active = qblk * 128 + wave * 32 < S;  // wave-uniform
  uint4 qf[4], ql[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    qf[ks] = *(const uint4*)(qbase + (row0 + qc) * ld + ks * 16 + lh * 8);
    if (PL == 2) ql[ks] = *(const uint4*)(qbase + plane + (row0 + qc) * ld + ks * 16 + lh * 8);
  }

  // staging: thread t loads 16 B (d chunk t&7) of K for keys (t>>3), (t>>3)+32 and of V for the
  // adjacent key pair 2*(t>>3), 2*(t>>3)+1 (adjacent keys stay adjacent under vt_pos).  Tile T holds keys 1 + 64 T ...
  const int kc = tid & 7, kr = tid >> 3;
  u32x4_t rk[2 * PL], rv[2 * PL];  // [plane][i]
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};
#define ATT_LOAD_KV(T)                                                                        \
  do {                                                                                        \
    _Pragma("unroll") for (int pl = 0; pl < PL; ++pl) {                                       \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                         \
        const int key = 1 + (T) * ATT_KT + kr + 32 * i;                                       \
        const int vkey = 1 + (T) * ATT_KT + 2 * kr + i;                                       \
        u32x4_t k4 = zero4, v4 = zero4;                                                       \
        if (key < S) k4 = *(const u32x4_t*)(kbase + pl * plane + (row0 + key) * ld + kc * 8); \
        if (vkey < S) v4 = *(const u32x4_t*)(vbase + pl * plane + (row0 + vkey) * ld + kc * 8); \
        rk[pl * 2 + i] = k4;                                                                  \
        rv[pl * 2 + i] = v4;                                                                  \
      }                                                                                       \
    }                                                                                         \
  } while (0)
#define ATT_STORE_KV(BUF)                                                                     \
  do {                                                                                        \
    _Pragma("unroll") for (int pl = 0; pl < PL; ++pl) {                                       \
      char* sk = smem + (BUF) * STAGE + pl * ATT_STAGE;                                       \
      char* sv = sk + ATT_K_BYTES;                                                            \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                         \
        const int row = kr + 32 * i;                                                          \
        *(u32x4_t*)(sk + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4)) = rk[pl * 2 + i];        \
      }                                                                                       \
      const int key_l = 2 * kr;                                                               \
      const int pos = (key_l & 32) | vt_pos(key_l & 31);                                      \
      const u32x4_t w0 = rv[pl * 2], w1 = rv[pl * 2 + 1];                                     \
      _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                         \
        const uint32_t a = (e & 1) ? (w0[e >> 1] >> 16) : (w0[e >> 1] & 0xffffu);             \
        const uint32_t c = (e & 1) ? (w1[e >> 1] >> 16) : (w1[e >> 1] & 0xffffu);             \
        const int d = kc * 8 + e;                                                             \
        *(uint32_t*)(sv + d * 128 + ((((pos >> 3) ^ ((d >> 1) & 7) ^ ((d >> 4) & 3))) << 4) + (pos & 7) * 2) = a | (c << 16); \
      }                                                                                       \
    }                                                                                         \
  } while (0)

  const int ntiles = (S - 1 + ATT_KT - 1) / ATT_KT;  // tiles over keys 1 .. S-1
  if (ntiles > 0) ATT_LOAD_KV(0);

  const float cexp = 0.125f * 1.4426950408889634f;  // softmax scale folded into exp2
  // ---- key 0 on the VALU: s0 = <q, k0> (this lane holds 32 of the 64 d of its query; the other 32 sit in lane ^ 32)
  f32x16_t o[2];
  float m_run, l_run;
  {
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float qv[8], kv[8];
      u32x4_t qh = {qf[ks].x, qf[ks].y, qf[ks].z, qf[ks].w}, qlo = zero4;
      if (PL == 2) qlo = u32x4_t{ql[ks].x, ql[ks].y, ql[ks].z, ql[ks].w};
      unpack8x<DT, PL>(qh, qlo, qv);
      const u32x4_t kh = *(const u32x4_t*)(kbase + row0 * ld + ks * 16 + lh * 8);
      u32x4_t klo = zero4;
      if (PL == 2) klo = *(const u32x4_t*)(kbase + plane + row0 * ld + ks * 16 + lh * 8);
      unpack8x<DT, PL>(kh, klo, kv);
#pragma unroll
      for (int e = 0; e < 8; ++e) part = fmaf(qv[e], kv[e], part);
    }
    m_run = part + __shfl_xor(part, 32, 64);
    l_run = lh == 0 ? 1.f : 0.f;  // p0 = exp2(0) = 1, counted once (the two halves' partial sums are added at the end)
    // O^T[d][q] = p0 * v0[d]: this lane's rows d = dt*32 + 8g + 4 lh + (0..3)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * lh;
        const uint2 vh = *(const uint2*)(vbase + row0 * ld + d);
        float v4[4] = {T16<DT>::tof((uint16_t)(vh.x & 0xffffu)), T16<DT>::tof((uint16_t)(vh.x >> 16)),
                       T16<DT>::tof((uint16_t)(vh.y & 0xffffu)), T16<DT>::tof((uint16_t)(vh.y >> 16))};
        if (PL == 2) {
          const uint2 vl = *(const uint2*)(vbase + plane + row0 * ld + d);
          v4[0] += T16<DT>::tof((uint16_t)(vl.x & 0xffffu)); v4[1] += T16<DT>::tof((uint16_t)(vl.x >> 16));
          v4[2] += T16<DT>::tof((uint16_t)(vl.y & 0xffffu)); v4[3] += T16<DT>::tof((uint16_t)(vl.y >> 16));
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) o[dt][4 * g + e] = v4[e];
      }
  }

  if (ntiles > 0) ATT_STORE_KV(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const bool more = (t + 1) < ntiles;
    if (more) ATT_LOAD_KV(t + 1);
    const char* sk = smem + (t & 1) * STAGE;
    const char* sv = sk + ATT_K_BYTES;
    if (wave_active) {
      // ---- S^T = K Q^T for both 32-key halves: two independent accumulators
      f32x16_t s0, s1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int chunk = 2 * ks + lh;
        const int koff0 = lr * 128 + ((chunk ^ ((lr >> 1) & 7)) << 4);  // rows lr and 32 + lr share (row >> 1) & 7
        const int koff1 = koff0 + 32 * 128;
        const uint4 kf0 = *(const uint4*)(sk + koff0);
        const uint4 kf1 = *(const uint4*)(sk + koff1);
        if (PL == 2) {
          const uint4 kl0 = *(const uint4*)(sk + ATT_STAGE + koff0);
          const uint4 kl1 = *(const uint4*)(sk + ATT_STAGE + koff1);
          s0 = T16<DT>::mfma32(kl0, qf[ks], s0);
          s1 = T16<DT>::mfma32(kl1, qf[ks], s1);
          s0 = T16<DT>::mfma32(kf0, ql[ks], s0);
          s1 = T16<DT>::mfma32(kf1, ql[ks], s1);
        }
        s0 = T16<DT>::mfma32(kf0, qf[ks], s0);
        s1 = T16<DT>::mfma32(kf1, qf[ks], s1);
      }
      // s0[r] / s1[r] = <K[key], Q[q]> with key = 1 + t*64 + {0, 32} + (r&3) + 8*(r>>2) + 4*lh, q = this lane's column
      const int key0 = 1 + t * ATT_KT + 4 * lh;
      if (key0 + 32 + 28 + 3 >= S) {  // only the last tile can hold keys >= S
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (key0 + (r & 3) + 8 * (r >> 2) >= S) s0[r] = -1e30f;
          if (key0 + 32 + (r & 3) + 8 * (r >> 2) >= S) s1[r] = -1e30f;
        }
      }
      float mx = s0[0], mx1 = s1[0];  // two chains (hipcc folds each into v_max3_f32)
#pragma unroll
      for (int r = 1; r < 16; ++r) { mx = fmaxf(mx, s0[r]); mx1 = fmaxf(mx1, s1[r]); }
      mx = fmaxf(mx, mx1);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cexp);  // raw v_exp_f32: args are <= 0
      const bool grew = m_new > m_run;
      m_run = m_new;
      const float mc = m_new * cexp;
      float pv0[16], pv1[16];
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        pv0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], cexp, -mc));
        pv1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], cexp, -mc));
        ps += pv0[r] + pv1[r];
      }
      l_run = l_run * alpha + ps;  // per-half partial sum; halves are added at the end
      if (__any(grew)) {           // wave-uniform: once the running max has settled the rescale is skipped
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
      }
      uint4 pf[4], pl2[4];  // [sub * 2 + s2]
      pf[0] = pack8<DT>(pv0);
      pf[1] = pack8<DT>(pv0 + 8);
      pf[2] = pack8<DT>(pv1);
      pf[3] = pack8<DT>(pv1 + 8);
      if (PL == 2) {
        float ph[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* src = (i < 2 ? pv0 : pv1) + (i & 1) * 8;
          unpack8<DT>(pf[i], ph);
#pragma unroll
          for (int r = 0; r < 8; ++r) ph[r] = src[r] - ph[r];
          pl2[i] = pack8<DT>(ph);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {  // i = sub * 2 + s2: the V^T chunk pair (2 i, 2 i + 1) holds these 16 keys
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const int drow = dt * 32 + lr;
          const int vchunk = i * 2 + lh;
          const int voff = drow * 128 + ((vchunk ^ ((drow >> 1) & 7) ^ ((drow >> 4) & 3)) << 4);
          const uint4 vf = *(const uint4*)(sv + voff);
          if (PL == 2) {
            const uint4 vl = *(const uint4*)(sv + ATT_STAGE + voff);
            o[dt] = T16<DT>::mfma32(vl, pf[i], o[dt]);
            o[dt] = T16<DT>::mfma32(vf, pl2[i], o[dt]);
          }
          o[dt] = T16<DT>::mfma32(vf, pf[i], o[dt]);
        }
      }
    }
    if (more) ATT_STORE_KV((t + 1) & 1);
    __syncthreads();
  }
#undef ATT_LOAD_KV
#undef ATT_STORE_KV

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  if (q < S) {
    uint16_t* op = out + (row0 + q) * (long long)(H * ATT_D) + head * ATT_D;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v0 = o[dt][4 * g] * inv, v1 = o[dt][4 * g + 1] * inv, v2 = o[dt][4 * g + 2] * inv, v3 = o[dt][4 * g + 3] * inv;
        uint2 w;
        w.x = T16<DT>::pack2(v0, v1);
        w.y = T16<DT>::pack2(v2, v3);
        *(uint2*)(op + dt * 32 + 8 * g + 4 * lh) = w;
        if (PL == 2) {
          uint2 l;
          l.x = T16<DT>::pack2(v0 - T16<DT>::tof((uint16_t)(w.x & 0xffffu)), v1 - T16<DT>::tof((uint16_t)(w.x >> 16)));
          l.y = T16<DT>::pack2(v2 - T16<DT>::tof((uint16_t)(w.y & 0xffffu)), v3 - T16<DT>::tof((uint16_t)(w.y >> 16)));
          *(uint2*)(op + plane + dt * 32 + 8 * g + 4 * lh) = l;
        }
      }
  }
}

hipError_t launch_attention(int mode, const void* qkv, void* out, int B, int S, int heads, Planes pl, hipStream_t stream) {
  const int BH = B * heads;
  dim3 grid(((S + 127) / 128) * BH);
  if (mode == MODE_BF16)
    hipLaunchKernelGGL((attention_kernel<DT_BF16, 1>), grid, dim3(256), 2 * ATT_STAGE, stream, (const uint16_t*)qkv, (uint16_t*)out, S, heads, BH, 0ll);
  else if (mode == MODE_FP16)
    hipLaunchKernelGGL((attention_kernel<DT_FP16, 1>), grid, dim3(256), 2 * ATT_STAGE, stream, (const uint16_t*)qkv, (uint16_t*)out, S, heads, BH, 0ll);
  else if (mode == MODE_BF16X3) {
    ensure_dyn_smem((const void*)attention_kernel<DT_BF16, 2>, 4 * ATT_STAGE);
    hipLaunchKernelGGL((attention_kernel<DT_BF16, 2>), grid, dim3(256), 4 * ATT_STAGE, stream, (const uint16_t*)qkv, (uint16_t*)out, S, heads, BH, pl.act);
  } else if (mode == MODE_FP16X3) {
    ensure_dyn_smem((const void*)attention_kernel<DT_FP16, 2>, 4 * ATT_STAGE);
    hipLaunchKernelGGL((attention_kernel<DT_FP16, 2>), grid, dim3(256), 4 * ATT_STAGE, stream, (const uint16_t*)qkv, (uint16_t*)out, S, heads, BH, pl.act);
  } else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace dptx
