// attention.hip -- fused softmax(Q K^T / sqrt(64)) V for the 12 ViT blocks (timm Attention,
// call site vit.py:150-151; N = 577 tokens, 12 heads x 64).  Scores never reach HBM.
//
// Input  qkv[B*S][3*H*64] 16-bit, feature index = which*H*64 + head*64 + dim (timm's
//        reshape(B,N,3,heads,64) packing); output out[B*S][H*64] 16-bit.
// Grid   1-D, ceil(S/128) * B * H blocks (q-block slowest: K/V sharers stay on one XCD); 4 waves x 32 query rows.
// Per 64-key tile (shared by the 4 waves through LDS; both operands travel by LDS-DMA, `buffer_load ... lds`: no staging
// registers, no ds_write, no VALU -- round 4, below):
//   K tile [64 keys][64 d]   row-major, 16-B chunks XOR-swizzled (same scheme as gemm.hip; the DMA image is lane-linear, so
//                            the swizzle goes on the SOURCE chunk)
//   V tile [64 keys][64 d]   row-major as well; the PV MFMA's A fragment (a d row, eight keys) is a COLUMN of it, read with
//                            ds_read_b64_tr_b16: a 16-lane group reads 4 keys x 16 d (lane l' supplies the address of
//                            V[key l'/4][d 4 (l' % 4) ..]) and lane l' receives V[4 keys][d l'] -- four consecutive keys of its
//                            d row; two reads give the eight keys in the order of the P^T registers (element e -> key
//                            (e & 3) + 8 (e >> 2) + 4 lh), so the P^T accumulators of the first MFMA are directly the B operand
//                            of the second one.  Position chunk p of row r holds source chunk p ^ 4 ((r >> 1) & 1): the eight
//                            rows one transposing read touches then fill two whole 256-byte bank rows.
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

__device__ __forceinline__ uint4 lds_read16(const char* p) {
  const u32x4_t v = *(const u32x4_t*)p;
  return uint4{v.x, v.y, v.z, v.w};
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
//
// Round 4 (tools/gpu/att_probe.py: the kernel with parts removed -- without the K / V staging 49 of 79 us, without the MFMAs 49,
// without exp / sum / max 56, skeleton 24): the staging -- global load -> registers -> VALU transposition of V -> ten ds_write per
// thread and tile -- was the largest single item.  K and V now travel by LDS-DMA and V is transposed by the fragment read
// (ds_read_b64_tr_b16): 79.4 -> 62.8 us at op level, 158 -> 112 registers (four blocks per CU instead of three), results
// bit-identical to the round-3 kernel.
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

  // staging: per tile and plane wave w issues pieces 2w, 2w+1 of K and of V (a piece = 8 keys x 128 B, lane-linear: lane l
  // writes chunk l & 7 of row l >> 3, so it fetches the source chunk that belongs there); keys >= S get an out-of-range buffer
  // offset, which the hardware returns as zeros.  Tile T holds keys 1 + 64 T ...
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};
#if defined(__HIP_DEVICE_COMPILE__)
  const int qkv_bytes = (int)((long long)(BH / H) * S * ld * 2);   // < 2^31 (launch_attention)
  const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(qkv), 0, qkv_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(qkv) + (PL == 2 ? plane : 0), 0, qkv_bytes, 0x00020000);
  auto dma_kv = [&](int T, int buf) {
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
      const int row = (wave * 2 + pc) * 8 + (lane >> 3);
      const int key = 1 + T * ATT_KT + row;
      const bool ok = key < S;
      const unsigned rowoff = (unsigned)((row0 + key) * ld + head * ATT_D);
      const unsigned koff = ok ? (rowoff + H * ATT_D + (((lane & 7) ^ ((row >> 1) & 7)) << 3)) * 2u : 0x80000000u;
      const unsigned voff = ok ? (rowoff + 2 * H * ATT_D + (((lane & 7) ^ (4 * ((row >> 1) & 1))) << 3)) * 2u : 0x80000000u;
      char* dst = smem + buf * STAGE + (wave * 2 + pc) * 1024;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc0, (__attribute__((address_space(3))) void*)dst, 16, koff, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc0, (__attribute__((address_space(3))) void*)(dst + ATT_K_BYTES), 16, voff, 0, 0, 0);
      if (PL == 2) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (__attribute__((address_space(3))) void*)(dst + ATT_STAGE), 16, koff, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc1, (__attribute__((address_space(3))) void*)(dst + ATT_STAGE + ATT_K_BYTES), 16, voff, 0, 0, 0);
      }
    }
  };
  // transposing V fragment reads: this lane's address inside the V tile for d block dt (the i-th 16-key block and the second
  // half of the eight keys are immediate offsets: + 2048 i, + 1024)
  const int lq = lane & 15, g16 = (lane >> 4) & 1;
  int va[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
    va[dt] = (4 * lh + (lq >> 2)) * 128 + (((dt ^ ((lq >> 3) & 1)) * 4 + 2 * g16 + ((lq & 3) >> 1)) << 4) + (lq & 1) * 8;
  typedef short s16x4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_p;
  auto read_vt = [&](const char* vtile, int dt, int i) -> uint4 {
    const __attribute__((address_space(3))) char* vb = (const __attribute__((address_space(3))) char*)vtile + va[dt] + i * 2048;
    const s16x4_t r0_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)vb);
    const s16x4_t r1_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(vb + 1024));
    const uint2 u0 = __builtin_bit_cast(uint2, r0_), u1 = __builtin_bit_cast(uint2, r1_);
    return uint4{u0.x, u0.y, u1.x, u1.y};
  };
#else
  auto dma_kv = [&](int, int) {};
  auto read_vt = [&](const char*, int, int) -> uint4 { return uint4{0u, 0u, 0u, 0u}; };
#endif

  const int ntiles = (S - 1 + ATT_KT - 1) / ATT_KT;  // tiles over keys 1 .. S-1
  if (ntiles > 0) dma_kv(0, 0);

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

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile 0 have landed
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const bool more = (t + 1) < ntiles;
    if (more) dma_kv(t + 1, (t + 1) & 1);   // the other stage: everybody left it at the barrier below
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
        // (ext-vector loads, not HIP's uint4 struct: behind a struct-typed LDS load hipcc waits vmcnt(0) for the LDS-DMA issued
        //  at the top of the iteration -- the prefetch of tile t + 1 would be drained before tile t is multiplied)
        const uint4 kf0 = lds_read16(sk + koff0), kf1 = lds_read16(sk + koff1);
        if (PL == 2) {
          const uint4 kl0 = lds_read16(sk + ATT_STAGE + koff0), kl1 = lds_read16(sk + ATT_STAGE + koff1);
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
          const uint4 vf = read_vt(sv, dt, i);
          if (PL == 2) {
            const uint4 vl = read_vt(sv + ATT_STAGE, dt, i);
            o[dt] = T16<DT>::mfma32(vl, pf[i], o[dt]);
            o[dt] = T16<DT>::mfma32(vf, pl2[i], o[dt]);
          }
          o[dt] = T16<DT>::mfma32(vf, pf[i], o[dt]);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile t + 1 have landed
    __syncthreads();
  }

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
  // 32-bit buffer offsets of the LDS-DMA.  Through the engine this cannot fire: dptx_create bounds max_batch * max_h * max_w * 256
  // below 2^31, which is stricter for both backbones (engine.hip dptx_create); it guards the op-level entry point (dptx_op_attention)
  if ((long long)B * S * 3 * heads * ATT_D * 2 >= (1ll << 31)) return hipErrorInvalidValue;
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
