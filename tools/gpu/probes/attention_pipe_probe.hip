// attention_pipe_probe.hip -- PROTOTYPE, not part of libdptx.so (round 4, for round 5).  The library's attention kernel (K / V by
// LDS-DMA, transposing V fragment reads) with the QK^T MFMAs of tile t + 1 issued BEFORE the softmax of tile t: the matrix pipe
// works under the wave's own VALU phase.  Costs a second pair of score accumulators (32 registers: three blocks per CU instead
// of four) and a third LDS stage (K of tile t + 1 must be resident one step earlier).  One plane (bf16 / fp16) only.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I omnidata_amd/csrc [-DATT_PIPE_BLOCKS=3] tools/gpu/probes/attention_pipe_probe.hip -o tools/gpu/probes/libattpipe[3].so
//   (2 blocks per CU: 205 registers, no scratch; 3: 168 registers, 132 bytes of scratch per lane)
//   python tools/gpu/att_probe.py pipe
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


#ifndef ATT_PIPE_BLOCKS
#define ATT_PIPE_BLOCKS 2
#endif
template <int DT>
__global__ __launch_bounds__(256, ATT_PIPE_BLOCKS) void attention_pipe_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out,
                                                                int S, int H, int BH) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 3 stages x (K tile + V tile)
  constexpr int STAGE = ATT_STAGE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int bh = (int)blockIdx.x % BH, qblk = (int)blockIdx.x / BH;
  const int head = bh % H, b = bh / H;
  const int ld = 3 * H * ATT_D;
  const long long row0 = (long long)b * S;
  const uint16_t* qbase = qkv + head * ATT_D;
  const uint16_t* kbase = qkv + H * ATT_D + head * ATT_D;
  const uint16_t* vbase = qkv + 2 * H * ATT_D + head * ATT_D;
  const int q = qblk * 128 + wave * 32 + lr;
  const int qc = q < S ? q : S - 1;
  const bool wave_active = qblk * 128 + wave * 32 < S;
  uint4 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const uint4*)(qbase + (row0 + qc) * ld + ks * 16 + lh * 8);
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};
  const int qkv_bytes = (int)((long long)(BH / H) * S * ld * 2);
  const __amdgpu_buffer_rsrc_t rsrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(qkv), 0, qkv_bytes, 0x00020000);
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
    }
  };
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
  const int ntiles = (S - 1 + ATT_KT - 1) / ATT_KT;
  if (ntiles > 0) dma_kv(0, 0);
  if (ntiles > 1) dma_kv(1, 1);
  const float cexp = 0.125f * 1.4426950408889634f;
  f32x16_t o[2];
  float m_run, l_run;
  {
    float part = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      float qv[8], kv[8];
      u32x4_t qh = {qf[ks].x, qf[ks].y, qf[ks].z, qf[ks].w};
      unpack8x<DT, 1>(qh, zero4, qv);
      const u32x4_t kh = *(const u32x4_t*)(kbase + row0 * ld + ks * 16 + lh * 8);
      unpack8x<DT, 1>(kh, zero4, kv);
#pragma unroll
      for (int e = 0; e < 8; ++e) part = fmaf(qv[e], kv[e], part);
    }
    m_run = part + __shfl_xor(part, 32, 64);
    l_run = lh == 0 ? 1.f : 0.f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * lh;
        const uint2 vh = *(const uint2*)(vbase + row0 * ld + d);
        o[dt][4 * g + 0] = T16<DT>::tof((uint16_t)(vh.x & 0xffffu)); o[dt][4 * g + 1] = T16<DT>::tof((uint16_t)(vh.x >> 16));
        o[dt][4 * g + 2] = T16<DT>::tof((uint16_t)(vh.y & 0xffffu)); o[dt][4 * g + 3] = T16<DT>::tof((uint16_t)(vh.y >> 16));
      }
  }
  // S^T of tile T (stage T % 3) into (s0, s1)
  auto qk = [&](int T, f32x16_t& s0, f32x16_t& s1) {
    const char* sk = smem + (T % 3) * STAGE;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int chunk = 2 * ks + lh;
      const int koff0 = lr * 128 + ((chunk ^ ((lr >> 1) & 7)) << 4);
      const uint4 kf0 = lds_read16(sk + koff0), kf1 = lds_read16(sk + koff0 + 32 * 128);
      s0 = T16<DT>::mfma32(kf0, qf[ks], s0);
      s1 = T16<DT>::mfma32(kf1, qf[ks], s1);
    }
  };
  // online softmax of tile T's scores and O += V^T P^T
  auto softmax_pv = [&](int T, f32x16_t& s0, f32x16_t& s1) {
    const char* sv = smem + (T % 3) * STAGE + ATT_K_BYTES;
    const int key0 = 1 + T * ATT_KT + 4 * lh;
    if (key0 + 32 + 28 + 3 >= S) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (key0 + (r & 3) + 8 * (r >> 2) >= S) s0[r] = -1e30f;
        if (key0 + 32 + (r & 3) + 8 * (r >> 2) >= S) s1[r] = -1e30f;
      }
    }
    float mx = s0[0], mx1 = s1[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) { mx = fmaxf(mx, s0[r]); mx1 = fmaxf(mx1, s1[r]); }
    mx = fmaxf(mx, mx1);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cexp);
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
    l_run = l_run * alpha + ps;
    if (__any(grew)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
    }
    uint4 pf[4];
    pf[0] = pack8<DT>(pv0); pf[1] = pack8<DT>(pv0 + 8); pf[2] = pack8<DT>(pv1); pf[3] = pack8<DT>(pv1 + 8);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) o[dt] = T16<DT>::mfma32(read_vt(sv, dt, i), pf[i], o[dt]);
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  f32x16_t a0, a1, b0, b1;
  if (wave_active && ntiles > 0) qk(0, a0, a1);
  for (int t = 0; t < ntiles; t += 2) {
    if (t + 2 < ntiles) dma_kv(t + 2, (t + 2) % 3);   // the stage of tile t - 1: everybody left it at the last barrier
    if (wave_active) {
      if (t + 1 < ntiles) qk(t + 1, b0, b1);           // the matrix pipe works on the next tile's scores ...
      softmax_pv(t, a0, a1);                            // ... under this tile's softmax
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t + 1 >= ntiles) break;
    if (t + 3 < ntiles) dma_kv(t + 3, (t + 3) % 3);
    if (wave_active) {
      if (t + 2 < ntiles) qk(t + 2, a0, a1);
      softmax_pv(t + 1, b0, b1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
        uint2 w;
        w.x = T16<DT>::pack2(o[dt][4 * g] * inv, o[dt][4 * g + 1] * inv);
        w.y = T16<DT>::pack2(o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        *(uint2*)(op + dt * 32 + 8 * g + 4 * lh) = w;
      }
  }
#endif
}

void ensure_dyn_smem(const void*, size_t) {}
}  // namespace dptx

extern "C" int att_pipe(const void* qkv, void* out, int B, int S, int heads, void* stream) {
  const int BH = B * heads;
  dim3 grid(((S + 127) / 128) * BH);
  static bool once = false;
  if (!once) { (void)hipFuncSetAttribute((const void*)dptx::attention_pipe_kernel<dptx::DT_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * dptx::ATT_STAGE); once = true; }
  hipLaunchKernelGGL((dptx::attention_pipe_kernel<dptx::DT_BF16>), grid, dim3(256), 3 * dptx::ATT_STAGE, (hipStream_t)stream, (const uint16_t*)qkv,
                     (uint16_t*)out, S, heads, BH);
  return (int)hipGetLastError();
}
