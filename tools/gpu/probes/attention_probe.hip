// attention_probe.hip -- TIMING PROBE, not part of libdptx.so.  A copy of omnidata_amd/csrc/attention.hip (round 4) whose kernel
// takes an ablation mask as template parameter, so that one run tells where the 67 us of a launch go:
//   1 no exponential (p = the fused multiply-add's result)   2 no running sum   4 no running max (m stays at s0)
//   8 no PV MFMAs   16 no QK MFMAs (scores = 0)   32 no K / V staging after the first tile (LDS keeps tile 0; barrier stays)
//   128 the K tile by LDS-DMA instead of load -> register -> ds_write (correct results)
//   256 K and V by LDS-DMA, V row-major, PV fragments by ds_read_b64_tr_b16 (correct results)
//   64 PACKED softmax arithmetic (v_pk_fma_f32 / v_pk_add_f32 on register pairs): results stay correct
// Every mask but 0 and 64 computes WRONG results.  Build (cross-compiles without a GPU):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I omnidata_amd/csrc tools/gpu/probes/attention_probe.hip -o tools/gpu/probes/libattprobe.so
// Run: python tools/gpu/att_probe.py
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
template <int DT, int PL, int ABL>
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
        if (!(ABL & 384) && key < S) k4 = *(const u32x4_t*)(kbase + pl * plane + (row0 + key) * ld + kc * 8); \
        if (!(ABL & 256) && vkey < S) v4 = *(const u32x4_t*)(vbase + pl * plane + (row0 + vkey) * ld + kc * 8); \
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
      if (!(ABL & 384)) {                                                                     \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                         \
        const int row = kr + 32 * i;                                                          \
        *(u32x4_t*)(sk + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4)) = rk[pl * 2 + i];        \
      }                                                                                       \
      }                                                                                       \
      if (ABL & 256) continue;                                                                \
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

  // ABL & 128: the K tile travels by LDS-DMA (buffer_load ... lds): wave w issues pieces 2w, 2w+1 (8 keys x 128 B each); the
  // image is lane-linear, so the XOR swizzle goes on the SOURCE chunk; keys >= S get an out-of-range offset (reads as zero)
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rsrcK = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(qkv), 0, (int)((long long)gridDim.x / ((S + 127) / 128) / H * S * ld * 2), 0x00020000);
  auto dma_k = [&](int T, int buf) {
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      const int row = (wave * 2 + pc) * 8 + (lane >> 3);
      const int key = 1 + T * ATT_KT + row;
      const int sc = (lane & 7) ^ ((row >> 1) & 7);
      const unsigned off = key < S ? (unsigned)(((row0 + key) * ld + H * ATT_D + head * ATT_D + sc * 8) * 2) : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcK, (__attribute__((address_space(3))) void*)(smem + buf * STAGE + (wave * 2 + pc) * 1024), 16, off, 0, 0, 0);
    }
  };
  // ABL & 256: V row-major [64 keys][128 B] by LDS-DMA too; position chunk p of row r holds source chunk p ^ 4 ((r >> 1) & 1), so
  // that the eight rows one ds_read_b64_tr_b16 touches (4 lh + 0..3) fill two whole 256-byte bank rows
  auto dma_v = [&](int T, int buf) {
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      const int row = (wave * 2 + pc) * 8 + (lane >> 3);
      const int key = 1 + T * ATT_KT + row;
      const int sc = (lane & 7) ^ (4 * ((row >> 1) & 1));
      const unsigned off = key < S ? (unsigned)(((row0 + key) * ld + 2 * H * ATT_D + head * ATT_D + sc * 8) * 2) : 0x80000000u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcK, (__attribute__((address_space(3))) void*)(smem + buf * STAGE + ATT_K_BYTES + (wave * 2 + pc) * 1024), 16, off, 0, 0, 0);
    }
  };
  // transposing fragment reads: lane (lh, g16, l') supplies the address of V[key 4 lh + l'/4][d 16 g16 + 4 (l' % 4) ...] and receives
  // V[4 lh + 0..3][16 g16 + l'] -- four consecutive keys of ITS d row; two reads = the eight keys of the MFMA's A fragment in the
  // key order of the P^T registers (e -> key (e & 3) + 8 (e >> 2) + 4 lh)
  const int lq = lane & 15, g16 = (lane >> 4) & 1;
  int va[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
    va[dt] = (4 * lh + (lq >> 2)) * 128 + (((dt ^ ((lq >> 3) & 1)) * 4 + 2 * g16 + ((lq & 3) >> 1)) << 4) + (lq & 1) * 8;
#else
  auto dma_k = [&](int, int) {};
  auto dma_v = [&](int, int) {};
  int va[2] = {0, 0};
#endif
  const int ntiles = (S - 1 + ATT_KT - 1) / ATT_KT;  // tiles over keys 1 .. S-1
  if (ntiles > 0) { ATT_LOAD_KV(0); if (ABL & 384) dma_k(0, 0); if (ABL & 256) dma_v(0, 0); }

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
  if (ABL & 384) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const bool more = (t + 1) < ntiles;
    if (more && !(ABL & 32)) { ATT_LOAD_KV(t + 1); if (ABL & 384) dma_k(t + 1, (t + 1) & 1); if (ABL & 256) dma_v(t + 1, (t + 1) & 1); }
    const char* sk = smem + ((ABL & 32) ? 0 : (t & 1)) * STAGE;
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
        if (!(ABL & 16)) {
          s0 = T16<DT>::mfma32(kf0, qf[ks], s0);
          s1 = T16<DT>::mfma32(kf1, qf[ks], s1);
        } else if (ks == 0) { s0[0] += __uint_as_float(kf0.x & 1u); s1[0] += __uint_as_float(kf1.x & 1u); }  // keep the reads alive
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
      if (!(ABL & 4)) {
#pragma unroll
        for (int r = 1; r < 16; ++r) { mx = fmaxf(mx, s0[r]); mx1 = fmaxf(mx1, s1[r]); }
        mx = fmaxf(mx, mx1);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      } else mx = m_run;
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cexp);  // raw v_exp_f32: args are <= 0
      const bool grew = m_new > m_run;
      m_run = m_new;
      const float mc = m_new * cexp;
      float pv0[16], pv1[16];
      float ps = 0.f;
      if (ABL & 64) {   // packed arithmetic on register pairs
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        const f32x2_t c2 = {cexp, cexp}, m2 = {-mc, -mc};
        f32x2_t acc2 = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          f32x2_t a = {s0[r], s0[r + 1]}, b = {s1[r], s1[r + 1]};
          a = __builtin_elementwise_fma(a, c2, m2);
          b = __builtin_elementwise_fma(b, c2, m2);
          f32x2_t pa = {__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)};
          f32x2_t pb = {__builtin_amdgcn_exp2f(b.x), __builtin_amdgcn_exp2f(b.y)};
          pv0[r] = pa.x; pv0[r + 1] = pa.y; pv1[r] = pb.x; pv1[r + 1] = pb.y;
          acc2 += pa + pb;
        }
        ps = acc2.x + acc2.y;
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a0 = fmaf(s0[r], cexp, -mc), a1 = fmaf(s1[r], cexp, -mc);
          pv0[r] = (ABL & 1) ? a0 : __builtin_amdgcn_exp2f(a0);
          pv1[r] = (ABL & 1) ? a1 : __builtin_amdgcn_exp2f(a1);
          if (!(ABL & 2)) ps += pv0[r] + pv1[r];
        }
        if (ABL & 2) ps = pv0[0];
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
          uint4 vf;
          if (ABL & 256) {
#if defined(__HIP_DEVICE_COMPILE__)
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) s16x4* lds_s16x4_p;
            const __attribute__((address_space(3))) char* vb = (const __attribute__((address_space(3))) char*)sv + va[dt] + i * 2048;
            const s16x4 r0_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(vb));
            const s16x4 r1_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(vb + 1024));
            const uint2 u0 = __builtin_bit_cast(uint2, r0_), u1 = __builtin_bit_cast(uint2, r1_);
            vf = uint4{u0.x, u0.y, u1.x, u1.y};
#endif
          } else vf = *(const uint4*)(sv + voff);
          if (PL == 2) {
            const uint4 vl = *(const uint4*)(sv + ATT_STAGE + voff);
            o[dt] = T16<DT>::mfma32(vl, pf[i], o[dt]);
            o[dt] = T16<DT>::mfma32(vf, pl2[i], o[dt]);
          }
          if (!(ABL & 8)) o[dt] = T16<DT>::mfma32(vf, pf[i], o[dt]);
          else if (i == 0) o[dt][0] += __uint_as_float((vf.x ^ pf[i].x) & 1u);   // keep the reads and the packing alive
          else o[dt][1] += __uint_as_float(pf[i].x & 1u);
        }
      }
    }
    if (more && !(ABL & 32)) ATT_STORE_KV((t + 1) & 1);
    if (ABL & 384) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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

}  // namespace dptx

namespace dptx { void ensure_dyn_smem(const void*, size_t) {} }

template <int ABL>
static int run(const void* qkv, void* out, int B, int S, int heads, hipStream_t stream) {
  const int BH = B * heads;
  dim3 grid(((S + 127) / 128) * BH);
  hipLaunchKernelGGL((dptx::attention_kernel<dptx::DT_BF16, 1, ABL>), grid, dim3(256), 2 * dptx::ATT_STAGE, stream, (const uint16_t*)qkv,
                     (uint16_t*)out, S, heads, BH, 0ll);
  return (int)hipGetLastError();
}

// ds_read_b64_tr_b16 semantics: LDS holds its own 16-bit index; lane l reads at byte address addr[l]; out[l][0..3]
__global__ void tr_probe_kernel(const int* __restrict__ addr, unsigned short* __restrict__ out) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
  u32x2v r;
  const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (unsigned)addr[threadIdx.x];
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  out[threadIdx.x * 4 + 0] = (unsigned short)(r.x & 0xffffu);
  out[threadIdx.x * 4 + 1] = (unsigned short)(r.x >> 16);
  out[threadIdx.x * 4 + 2] = (unsigned short)(r.y & 0xffffu);
  out[threadIdx.x * 4 + 3] = (unsigned short)(r.y >> 16);
#endif
}
extern "C" int tr_probe(const int* addr, void* out, void* stream) {
  hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, addr, (unsigned short*)out);
  return (int)hipGetLastError();
}

extern "C" int att_probe(int mask, const void* qkv, void* out, int B, int S, int heads, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  switch (mask) {
    case 0: return run<0>(qkv, out, B, S, heads, st);
    case 1: return run<1>(qkv, out, B, S, heads, st);
    case 2: return run<2>(qkv, out, B, S, heads, st);
    case 3: return run<3>(qkv, out, B, S, heads, st);
    case 4: return run<4>(qkv, out, B, S, heads, st);
    case 7: return run<7>(qkv, out, B, S, heads, st);
    case 8: return run<8>(qkv, out, B, S, heads, st);
    case 16: return run<16>(qkv, out, B, S, heads, st);
    case 24: return run<24>(qkv, out, B, S, heads, st);
    case 31: return run<31>(qkv, out, B, S, heads, st);
    case 32: return run<32>(qkv, out, B, S, heads, st);
    case 63: return run<63>(qkv, out, B, S, heads, st);
    case 64: return run<64>(qkv, out, B, S, heads, st);
    case 128: return run<128>(qkv, out, B, S, heads, st);
    case 192: return run<192>(qkv, out, B, S, heads, st);
    case 256: return run<256>(qkv, out, B, S, heads, st);
    case 320: return run<320>(qkv, out, B, S, heads, st);
  }
  return -1;
}
