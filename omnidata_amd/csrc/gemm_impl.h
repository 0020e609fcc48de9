#pragma once
// Included by gemm.hip (launch_gemm + bf16), gemm_fp16.hip, gemm_x3.hip (hi/lo-plane 3-MFMA modes) and gemm_fp8.hip
// (e4m3): four translation units, so that hipcc builds the instantiation sets in parallel.
// gemm_impl.h -- implicit-GEMM MFMA kernels for every linear / 1x1 / 3x3 / 7x7(im2col'd) layer of
// DPT-Hybrid (SURVEY.md A.6 lists the 44 unique shapes).  gfx950 only.
//
//   C[M,N] = epilogue( gatherA[M,K] * W[N,K]^T )
//
// * v_mfma_f32_32x32x16_{bf16,f16}: a wave owns a (TM*32) x (TN*32) accumulator block.
// * BK = 64: one k-tile of A / W is [rows][64] 16-bit = 128 B per row in LDS; the 16-B chunks of a
//   row are XOR-swizzled with ((row>>1)&7), which makes the ds_read_b128 fragment reads (16
//   distinct rows per lane group, same k chunk) bank-conflict free.
// * gemm_glds_kernel (default): both operands stream HBM/L2 -> LDS with `buffer_load_dwordx4 ...
//   lds` (no VGPR round trip, no ds_write).  The LDS image of a wave-instruction is lane-linear
//   (8 rows x 128 B), so the swizzle is applied to the per-lane SOURCE chunk; out-of-image conv
//   taps and rows >= M use an out-of-range buffer offset, which the hardware returns as zeros.
//   Two LDS stages: the loads of tile t+1 are in flight while tile t is multiplied; one
//   vmcnt(0)+barrier per k-tile.
// * gemm_reg_kernel: register-staged variant of the same tiling, used when A is fp32 (the
//   ProjectReadout GEMM reads the fp32 token stream and rounds while staging) or when the A
//   buffer is too large for a 32-bit buffer offset.
// * A is gathered as NHWC conv taps, so dense GEMM, strided 1x1 and kxk convolutions share the
//   loader; optional ReLU on the A fragments (RCU pre-activation, one v_pk_max_i16 per dword).
// * epilogue: accumulators -> LDS fp32 tile -> coalesced 16-B rows with fused bias, ReLU /
//   erf-GELU, up to two residuals (16-bit or fp32, one may broadcast over images), 16-bit or
//   fp32 output, optional row remap (token rows skip the cls slot).
// * 1-D grid, XCD-aware bijective remap; n-tile fastest so the blocks of one XCD re-use the
//   same A rows out of that XCD's L2.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace dptx {

constexpr int BK = 64;

// q = m / d, rem = m % d for 0 <= m < 2^23 without the ~35-instruction integer division: float(m) is exact, the product
// with the rounded reciprocal is off by < 1, one correction step either way.  `big` (uniform): exact division.
__device__ __forceinline__ int row_div(int m, int d, float rcp, bool big, int& rem) {
  if (big) {
    const int q = m / d;
    rem = m - q * d;
    return q;
  }
  int q = (int)((float)m * rcp);
  int r = m - q * d;
  if (r < 0) { --q; r += d; }
  if (r >= d) { ++q; r -= d; }
  rem = r;
  return q;
}

// (mu, rstd) of a token row from its (sum, sum of squares): var = E[x^2] - mu^2 in double -- the subtraction is where the bits go
__device__ __forceinline__ float2 ln_mu_rstd(float sm, float sq, float inv_dim, float eps) {
  const float mu = sm * inv_dim;
  const float var = (float)((double)sq * (double)inv_dim - (double)mu * (double)mu);
  return make_float2(mu, __builtin_amdgcn_rsqf(fmaxf(var, 0.f) + eps));
}

// LayerNorm fold, consumer side: y = (acc - mu * colsum) * rstd + bias as two fused multiply-adds -- every kernel form that can
// serve a layer rounds identically -- and as SCALAR v_fma_f32.  Left to itself hipcc pairs the elements into
// v_pk_fma_f32 ... op_sel:[0,1,0]: both lanes of the packed op take rstd from the HIGH dword of the (mu, rstd) register pair
// the table entry was loaded into.  On MI355X that form returned `bias` -- a product of zero -- in the LOW lane for work-items
// 48..63 of a wave, a few times per thousand launches, whenever the stem convolution of ANOTHER stream shared the CU
// (tools/gpu/r4_micro.py reproduces it in seconds: 224-445 of 2400 launches with the packed form, 0 of 2400 with this one;
// operands recorded in the kernel were right).  It was round 3's "nondeterminism" of the multi-stream forward.
// The empty asm keeps the SLP vectoriser from pairing the elements; tests/test_build_quality.py checks the compiled ISA for
// packed fp32 operations whose low lane reads a high dword.
__device__ __forceinline__ float ln_fold_fma(float acc, float mu, float rstd, float colsum, float bias) {
  float t = fmaf(-mu, colsum, acc);
  asm volatile("" : "+v"(t));
  t = fmaf(t, rstd, bias);
  asm volatile("" : "+v"(t));
  return t;
}

// ----------------------------------------------------------------------------- shared pieces
template <int DT, int TM, int TN, bool RELU_A, int PL = 1, int HK = BK / 16, int XT = 3>
__device__ __forceinline__ void mma_tile(const char* sa, const char* sb, int a_lo, int b_lo, int wm, int wn, int lr, int lh,
                                         f32x16_t (&acc)[TM][TN]) {
  // a_lo / b_lo: byte distance of the lo-plane tiles inside the stage (PL == 2)
  if constexpr (DT == DT_FP8) {
    // a row of the k-tile is 128 e4m3 bytes.  v_mfma_scale_f32_32x32x64_f8f6f4 (format 0 = e4m3 for both operands, E8M0
    // scale 127 = 1.0 in every byte of the scale registers: a plain fp8 MFMA at twice the bf16 rate) contracts 64 k per
    // instruction; lane (lr, lh) supplies 32 consecutive bytes of its row -- the two 16-B chunks 4q + 2lh, 4q + 2lh + 1
    // of MFMA q = 0, 1.  A and W use the same byte -> k assignment, so the contraction over the 128 bytes is complete
    // whatever k order the instruction uses internally.
    static_assert(PL == 1 && !RELU_A, "fp8: single plane; the producer writes the ReLU'd copy");
    typedef int i32x8_t __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      i32x8_t af[TM], bf[TN];
      const int c0 = 4 * q + 2 * lh;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (TM * 32) + i * 32 + lr;
        const int sw = (row >> 1) & 7;
        const u32x4_t lo = *(const u32x4_t*)(sa + row * 128 + ((c0 ^ sw) << 4));
        const u32x4_t hi = *(const u32x4_t*)(sa + row * 128 + (((c0 + 1) ^ sw) << 4));
        af[i] = i32x8_t{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * (TN * 32) + j * 32 + lr;
        const int sw = (row >> 1) & 7;
        const u32x4_t lo = *(const u32x4_t*)(sb + row * 128 + ((c0 ^ sw) << 4));
        const u32x4_t hi = *(const u32x4_t*)(sb + row * 128 + (((c0 + 1) ^ sw) << 4));
        bf[j] = i32x8_t{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[i], bf[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    return;
  } else if constexpr (PL == 1) {
    // all fragment reads of the k-tile are issued up front (16 ds_read_b128 in flight for a 64x64 wave tile,
    // 64 VGPRs) and the MFMAs consume them behind counted lgkmcnt waits: the LDS latency is paid once per
    // k-tile instead of once per k-step
    // (HK k-steps per group: 4 = the whole k-tile for 64x64 wave tiles; 2 for the 128x64 wave tile of the
    // 256x256 block, whose 128 accumulator registers leave room for 48 fragment registers, not 96)
#pragma unroll
    for (int g = 0; g < BK / 16; g += HK) {
      u32x4_t af[HK][TM], bf[HK][TN];
#pragma unroll
      for (int ks = 0; ks < HK; ++ks) {
        const int chunk = 2 * (g + ks) + lh;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wm * (TM * 32) + i * 32 + lr;
          af[ks][i] = *(const u32x4_t*)(sa + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = wn * (TN * 32) + j * 32 + lr;
          bf[ks][j] = *(const u32x4_t*)(sb + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
        }
      }
      __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMAs (the scheduler otherwise sinks them
                                          // back to one k-step of look-ahead to save registers)
#pragma unroll
      for (int ks = 0; ks < HK; ++ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if (RELU_A) af[ks][i] = relu8(af[ks][i]);
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = T16<DT>::mfma32(af[ks][i], bf[ks][j], acc[i][j]);
        }
      }
      if (HK != BK / 16) __builtin_amdgcn_sched_barrier(0);
    }
    return;
  } else {
    // hi/lo planes, 3 MFMAs per product.  The fragment reads run one k-step ahead of the MFMAs (two fragment sets): the
    // 2 (TM + TN) ds_read_b128 of k-step ks+1 are in flight under the 3 TM TN MFMAs of k-step ks
    // (DPTX_X3_NOPIPE: read, then multiply, per k-step -- the round-1 form, for A/B runs)
    // XT == 2 (GemmParams::a_hi_only): the activations contribute their hi plane only -- a_hi w_lo + a_hi w_hi, two MFMAs;
    // the A lo tile is neither loaded nor read
    u32x4_t af[2][TM], bf[2][TN], al[2][TM], bl[2][TN];
    auto read = [&](int s_, int ks) {
      const int chunk = 2 * ks + lh;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (TM * 32) + i * 32 + lr;
        const int off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
        af[s_][i] = *(const u32x4_t*)(sa + off);
        if (XT == 3) al[s_][i] = *(const u32x4_t*)(sa + a_lo + off);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * (TN * 32) + j * 32 + lr;
        const int off = row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
        bf[s_][j] = *(const u32x4_t*)(sb + off);
        bl[s_][j] = *(const u32x4_t*)(sb + b_lo + off);
      }
    };
    auto mma = [&](int s_) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (RELU_A) {
          if (XT == 3) relu8_planes(af[s_][i], al[s_][i]);
          else af[s_][i] = relu8(af[s_][i]);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {  // small cross terms first, then the leading term
          if (XT == 3) acc[i][j] = T16<DT>::mfma32(al[s_][i], bf[s_][j], acc[i][j]);
          acc[i][j] = T16<DT>::mfma32(af[s_][i], bl[s_][j], acc[i][j]);
          acc[i][j] = T16<DT>::mfma32(af[s_][i], bf[s_][j], acc[i][j]);
        }
      }
    };
#ifdef DPTX_X3_NOPIPE
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) { read(0, ks); mma(0); }
#else
    read(0, 0);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      if (ks + 1 < BK / 16) read((ks + 1) & 1, ks + 1);
      __builtin_amdgcn_sched_barrier(0);
      mma(ks & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
  }
}

// SLABS == 1: the whole BM x BN tile goes through LDS at once.  SLABS == TM (256x256 block: the fp32 tile would be
// 266 KB): TM passes, pass s carries the s-th 32-row MFMA tile of every wave -- LDS row q = (wave row)*32 + r is tile
// row (q/32)*(TM*32) + s*32 + q%32.
//
// Memory-level parallelism: a thread owns ITER rows x 8 columns of a slab.  All of its residual / per-image-bias loads
// are issued BEFORE the accumulators are staged through LDS, so their latency (1-2 us next to a second busy block) is
// paid once per slab under the staging, not once per row; with C == R1 (the in-place residual stream) every thread
// reads exactly the elements it later writes, so the order loads -> stores is also what makes that legal.
//
// GroupNorm statistics (p.gn_part): per 32-row MFMA block and group, (sum, sum of squares) of the fp32 accumulators,
// reduced in a fixed order (16 registers, lane^32, then lane^1..cpg/2) and written -- not accumulated -- to
// partial[img][block][group]: no atomics, and the value of a record depends on the image's own rows only, so the
// statistics are bit-identical run to run and at every batch size (needs rows-per-image % 32 == 0; the engine falls
// back to the gn_stats kernel otherwise).
//
// ILV (gemm_ph_kernel, 256x256, 2 x 4 waves): a wave's 4 x 2 MFMA blocks are interleaved over the tile -- row block i of
// wave row wm sits at tile row (i>>1)*128 + wm*64 + (i&1)*32, column block j of wave column wn at j*128 + wn*32 -- so that
// the four 128-row half-tiles of a k-tile are needed one phase after the other.
//
// WP (wave-private staging; gemm_pp_kernel): every wave transposes its own 32 x 64 block of a slab through its own 9 KB of
// LDS and writes its own 32 rows x 128 bytes -- no block barrier inside the epilogue, so one wave's LDS round trip runs
// under another's arithmetic and stores instead of all eight waves marching through write / barrier / read / compute /
// store in step (a plain bias epilogue took 17.6 k cycles per tile that way for ~4 k cycles of LDS time and ~2.5 k of
// VALU time: profiles/r03_experiments.md).  Not with row_stats (a record sums 128 columns: two waves).
template <int DT, int BM, int BN, int TM, int TN, int PL = 1, int NT = 256, int SLABS = 1, bool ILV = false, bool WP = false>
__device__ __forceinline__ void epilogue(const GemmParams& p, char* smem, int m0, int n0, int wm, int wn, int lr, int lh,
                                         int tid, f32x16_t (&acc)[TM][TN], int row_pitch = 0, bool dma_in_flight = false,
                                         int trace_row = -1, const char* lds_ln = nullptr) {
  // lds_ln (persistent gemm_pp_kernel): the (sum, sum of squares) records of the tile's BM rows, 64 bytes per tile row,
  // already in LDS (DMA'd at the start of the tile's k-loop; rows >= M read as zeros and are masked below)
  // dma_in_flight (persistent gemm_pp_kernel): the caller has LDS-DMA of its next tile outstanding; the epilogue waits for
  // it together with its own first loads, BEFORE its first store (stores count in vmcnt on gfx9: a vmcnt(0) after them
  // would wait for the tile to reach memory)
  // row_pitch != 0 (gemm_halo_kernel): the tile is 8 image rows x 32 pixels -- tile row R is GEMM row
  // m0 + (R >> 5) * row_pitch + (R & 31)
  static_assert(SLABS == 1 || SLABS == TM, "one slab, or one per MFMA row tile");
  static_assert(!ILV || (SLABS == TM && TM == 4 && TN == 2 && BM == 256 && BN == 256), "interleaved mapping: the phased kernel");
  static_assert(!WP || (SLABS == TM && TN == 2 && !ILV), "wave-private staging: one 32 x 64 block per wave and slab");
  constexpr int CT_PITCH = WP ? 72 : BN + 4;  // floats (WP: 4 rows apart = 32 banks apart)
  constexpr int CT_ROWS = WP ? 32 : BM / SLABS;
  constexpr int NCH = WP ? 8 : BN / 8;       // 8-column chunks per staged row
  constexpr int RPP = WP ? 8 : NT / NCH;     // staged rows per pass
  constexpr int ITER = CT_ROWS / RPP;
  constexpr int WAVES_N = BN / (TN * 32);
  static_assert(CT_ROWS % RPP == 0, "rows per pass must divide the slab");
#ifdef DPTX_TRACE   // tile-phase stamps of thread 0 of block 0 (tools/gpu/pp_trace.py): rows 48.. of wave 0's trace block
#define DPTX_ESTAMP(SLOT)                                                                                          \
  do {                                                                                                             \
    if (p.trace != nullptr && blockIdx.x == 0 && tid == 0 && trace_row >= 0)                                       \
      p.trace[(48 + trace_row) * 4 + (SLOT)] = (long long)__builtin_readcyclecounter();                            \
  } while (0)
#else
#define DPTX_ESTAMP(SLOT) do { } while (0)
  (void)trace_row;
#endif
  DPTX_ESTAMP(0);
  float* ct = (float*)smem + (WP ? (wm * WAVES_N + wn) * (32 * CT_PITCH) : 0);
  const int cn = WP ? tid & 7 : tid % NCH;
  const int rr = WP ? (tid & 63) >> 3 : tid / NCH;
  const int n = n0 + (WP ? wn * (TN * 32) : 0) + cn * 8;

  if (p.gn_part != nullptr) {
    const int cpg = p.gn_cpg;  // channels per group: 2..32, a power of two
    const int cpg_sh = __builtin_ctz(cpg);
    int gn_hw = p.gn_hw;
    asm volatile("" : "+s"(gn_hw));  // opaque, like c_rpi below: no hoisted divider constants in a tile loop
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      // first GEMM row of this wave's i-th 32-row block
      const int mb = m0 + (ILV ? (i >> 1) * 128 + wm * 64 + (i & 1) * 32 : (wm * TM + i) * 32);
      const int img = mb / gn_hw;
      const int blk = (mb - img * gn_hw) >> 5;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; sm += v; sq = fmaf(v, v, sq); }
        sm += __shfl_xor(sm, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        for (int o = 1; o < cpg; o <<= 1) { sm += __shfl_xor(sm, o, 64); sq += __shfl_xor(sq, o, 64); }
        if (lh == 0 && (lr & (cpg - 1)) == 0 && mb < p.M) {
          const int g = (n0 + (ILV ? j * 128 + wn * 32 : (wn * TN + j) * 32) + lr) >> cpg_sh;
          float2* dst = (float2*)p.gn_part + ((long long)img * p.gn_blocks + blk) * 32 + g;
          *dst = make_float2(sm, sq);
        }
      }
    }
  }

  float bias_c[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bias_c[e] = 0.f;
  if (p.bias != nullptr && !p.bias_per_img) {
    const float4 b0 = *(const float4*)(p.bias + n), b1 = *(const float4*)(p.bias + n + 4);
    bias_c[0] = b0.x; bias_c[1] = b0.y; bias_c[2] = b0.z; bias_c[3] = b0.w;
    bias_c[4] = b1.x; bias_c[5] = b1.y; bias_c[6] = b1.z; bias_c[7] = b1.w;
  }
  float osc[8];  // fp8: output scale of this thread's 8 columns
  if (p.out_scale != 0.f) {
#pragma unroll
    for (int e = 0; e < 8; ++e) osc[e] = p.out_scale;
    if (p.out_scale_v != nullptr) {
      const float4 s0 = *(const float4*)(p.out_scale_v + n), s1 = *(const float4*)(p.out_scale_v + n + 4);
      osc[0] *= s0.x; osc[1] *= s0.y; osc[2] *= s0.z; osc[3] *= s0.w;
      osc[4] *= s1.x; osc[5] *= s1.y; osc[6] *= s1.z; osc[7] *= s1.w;
    }
  }
  const bool remap = p.c_rpi != 0x7fffffff;  // token-row remap (patch-embed, readout); everything else skips the division
  int c_rpi = p.c_rpi;
  asm volatile("" : "+s"(c_rpi));  // opaque: inside a tile loop the divider's reciprocal would be hoisted into live registers
  // LayerNorm fold, consumer side: (mu, rstd) of the tile's BM rows, combined ONCE per row from its (sum, sum of squares)
  // records (row stride 8 records, ln_nblk = 6 or 8 valid) into an LDS table behind the C tile -- every row is needed by
  // the 32 (16) threads that own its columns, in every n-tile: per-thread combining cost 64 sixteen-byte loads per thread
  // and tile (+11..14 us per qkv / fc1 launch, profiles/r03_experiments.md).  The table is written before the first
  // __syncthreads() of the slab loop and read after it.
  float2* lnrow = (float2*)(smem + (WP ? (size_t)(NT / 64) * 32 * CT_PITCH * 4 : (size_t)CT_ROWS * CT_PITCH * 4));
  if (p.ln_stats != nullptr) {
    const bool all8 = p.ln_nblk == 8;
    for (int r = tid; r < BM; r += NT) {
      int R = r;
      if (row_pitch) R = (R >> 5) * row_pitch + (R & 31);
      int m = m0 + R;
      m = m < p.M ? m : m0;
      const float4* st = lds_ln != nullptr ? (const float4*)(lds_ln + r * 64) : (const float4*)(p.ln_stats + (long long)m * 16);
      const float4 r0 = st[0], r1_ = st[1], r2_ = st[2], r3_ = st[3];
      const float sm = (r0.x + r0.z) + (r1_.x + r1_.z) + ((r2_.x + r2_.z) + (all8 ? r3_.x + r3_.z : 0.f));
      const float sq = (r0.y + r0.w) + (r1_.y + r1_.w) + ((r2_.y + r2_.w) + (all8 ? r3_.y + r3_.w : 0.f));
      lnrow[r] = ln_mu_rstd(sm, sq, p.ln_inv_dim, p.ln_eps);
    }
    if (WP) __syncthreads();  // the only block barrier of the wave-private epilogue
  }
  DPTX_ESTAMP(1);
  // rows per load group: 4 (96 registers of loads in flight at most); 2 for the slab epilogue, whose 128 accumulator
  // registers stay live across the slabs
  constexpr int GR_ = SLABS > 1 ? 2 : (ITER < 4 ? ITER : 4);
  static_assert(ITER % GR_ == 0, "row groups");

  // The body is instantiated per residual configuration (R1M / R2M: 0 none, 1 16-bit, 2 fp32, -1 decided at run time;
  // BPI: per-image bias 0 / 1 / -1) and selected by ONE uniform branch below: inside an instance the loads are
  // unconditional, so the compiler keeps them where they are written -- all up front -- instead of sinking each one
  // into the conditional block that consumes it (which serialises a memory latency per row).
  auto body = [&](auto r1m_, auto r2m_, auto bpi_, auto lnf_) {
    constexpr int R1M = decltype(r1m_)::value, R2M = decltype(r2m_)::value, BPI = decltype(bpi_)::value;
    constexpr bool LNF = decltype(lnf_)::value != 0;  // LayerNorm folded into this GEMM (consumer side, kernels.h)
    // the run-time-decided instance of the slab epilogue (patch-embed, readout: five launches per forward) holds three load
    // sets per row: one row per group, or it spills inside the persistent tile loop of gemm_pp_kernel
    // (likewise the two-plane epilogue with two residuals: four 16-byte loads per row)
    constexpr int GR = (SLABS > 1 && (R1M < 0 || (PL == 2 && R1M != 0 && R2M != 0))) ? 1 : GR_;
    constexpr int NGR = ITER / GR;
    float ln_c[8];  // colsum of the folded weight for this thread's 8 columns
    if (LNF) {
      const float4 c0 = *(const float4*)(p.ln_colsum + n), c1 = *(const float4*)(p.ln_colsum + n + 4);
      ln_c[0] = c0.x; ln_c[1] = c0.y; ln_c[2] = c0.z; ln_c[3] = c0.w;
      ln_c[4] = c1.x; ln_c[5] = c1.y; ln_c[6] = c1.z; ln_c[7] = c1.w;
    }
    int ln_R[GR];  // tile row of the thread's row `it` (index into the LDS table of (mu, rstd))
    const bool r1 = R1M < 0 ? p.R1 != nullptr : R1M > 0, r2 = R2M < 0 ? p.R2 != nullptr : R2M > 0;
    const bool r1f = R1M < 0 ? p.r1_fp32 != 0 : R1M == 2, r2f = R2M < 0 ? p.r2_fp32 != 0 : R2M == 2;
    const bool bpi = BPI < 0 ? p.bias_per_img != 0 : BPI > 0;
    long long coff[GR];
    int crow_[GR];  // C row (< 2^31) of the LayerNorm-fold statistics record
    bool ok[GR];
    u32x4_t ra[GR][2], rb[GR][2], bb[GR][2];
    // addresses and every global load of row group `gi` of slab `s`
    auto issue_loads = [&](int s, int gi) {
#pragma unroll
      for (int it = 0; it < GR; ++it) {
        const int row = rr + (gi * GR + it) * RPP;
        int R = WP           ? wm * (TM * 32) + s * 32 + row
                : SLABS == 1 ? row
                : ILV        ? (s >> 1) * 128 + (row >> 5) * 64 + (s & 1) * 32 + (row & 31)
                             : (row >> 5) * (TM * 32) + s * 32 + (row & 31);
        if (LNF) ln_R[it] = R;  // tile row: index into the LDS table of (mu, rstd)
        if (row_pitch) R = (R >> 5) * row_pitch + (R & 31);
        int m = m0 + R;
        ok[it] = m < p.M;
        m = ok[it] ? m : m0;  // any valid row: the loads stay in bounds, the store is masked
        int img = 0, pp = m;
        if (remap) {
          img = m / c_rpi;
          pp = m - img * c_rpi;
        }
        const long long crow = (long long)img * p.c_img_rows + p.c_row_off + pp;
        coff[it] = crow * p.ldc + n;
        crow_[it] = (int)crow;
        if (r1) {
          if (r1f) {
            const u32x4_t* src = (const u32x4_t*)((const float*)p.R1 + coff[it]);
            ra[it][0] = src[0];
            ra[it][1] = src[1];
          } else {
            ra[it][0] = *(const u32x4_t*)((const uint16_t*)p.R1 + coff[it]);
            if (PL == 2) {
              ra[it][1] = u32x4_t{0u, 0u, 0u, 0u};
              if (!p.r1_hi_only) ra[it][1] = *(const u32x4_t*)((const uint16_t*)p.R1 + p.planes.act + coff[it]);
            }
          }
        }
        if (r2) {
          const long long off = p.r2_bcast ? (long long)(p.c_row_off + pp) * p.ldc + n : coff[it];
          if (r2f) {
            const u32x4_t* src = (const u32x4_t*)((const float*)p.R2 + off);
            rb[it][0] = src[0];
            rb[it][1] = src[1];
          } else {
            rb[it][0] = *(const u32x4_t*)((const uint16_t*)p.R2 + off);
            if (PL == 2) {
              rb[it][1] = u32x4_t{0u, 0u, 0u, 0u};
              if (!p.r2_hi_only) rb[it][1] = *(const u32x4_t*)((const uint16_t*)p.R2 + p.planes.act + off);
            }
          }
        }
        if (bpi) {
          const u32x4_t* src = (const u32x4_t*)(p.bias + (long long)img * p.N + n);
          bb[it][0] = src[0];
          bb[it][1] = src[1];
        }
      }
    };
#pragma unroll
    for (int s = 0; s < SLABS; ++s) {
      // ---- (a) the first row group's loads fly while the accumulators are staged
      issue_loads(s, 0);
      // ---- (b) accumulators -> LDS
      // WP: the wave's LDS accesses execute in order, the previous slab's reads are ahead of these writes in its queue
      if (WP) __builtin_amdgcn_wave_barrier();
      else if (s > 0) __syncthreads();  // the previous slab has been read out
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (SLABS != 1 && i != s) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int ml = (WP ? 0 : SLABS == 1 ? wm * (TM * 32) + i * 32 : wm * 32) + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int nl = (WP ? j * 32 : ILV ? j * 128 + wn * 32 : wn * (TN * 32) + j * 32) + lr;
            ct[ml * CT_PITCH + nl] = acc[i][j][r];
          }
      }
      if (s == 0 && dma_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (WP) __builtin_amdgcn_wave_barrier();
      else __syncthreads();
      // ---- (c) rows out
#pragma unroll
      for (int gi = 0; gi < NGR; ++gi) {
        if (gi > 0) issue_loads(s, gi);
#pragma unroll
        for (int it = 0; it < GR; ++it) {
          const int row = rr + (gi * GR + it) * RPP;
          float v[8];
          {
            const float4 x0 = *(const float4*)(ct + row * CT_PITCH + cn * 8);
            const float4 x1 = *(const float4*)(ct + row * CT_PITCH + cn * 8 + 4);
            v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w;
            v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
          }
          if (p.out_scale != 0.f) {  // fp8 GEMMs: weights (per output channel) and activations were scaled by powers of two
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= osc[e];
          }
          if (LNF) {
            // y = (acc - mu * colsum) * rstd + bias' as two EXPLICIT fused multiply-adds: every kernel form that can serve a
            // layer (this one, epilogue_direct, any tile shape) must round identically -- left to -ffp-contract the
            // compiler fuses "x * rstd + bias" at one call site and not at another (1-ulp differences, which the
            // batch-invariance test caught)
            const float2 ms = lnrow[ln_R[it]];  // (mu, rstd) of this row: one LDS broadcast read per 32 threads
#ifdef DPTX_LN_PACKED_FMA   /* round 3's form, for tools/gpu/r4_micro.py: hipcc turns this into v_pk_fma_f32 ... op_sel:[0,1,0] */
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(fmaf(-ms.x, ln_c[e], v[e]), ms.y, bias_c[e]);
#else
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = ln_fold_fma(v[e], ms.x, ms.y, ln_c[e], bias_c[e]);
#endif
          } else if (bpi) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] += __uint_as_float(bb[it][0][e]); v[4 + e] += __uint_as_float(bb[it][1][e]); }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias_c[e];
          }
          if (p.act == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          } else if (p.act == 2) {
#pragma unroll
            for (int e = 0; e < 8; e += 2) gelu_erf2(v[e], v[e + 1]);
          }
          if (r1) {
            if (r1f) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[e] += __uint_as_float(ra[it][0][e]); v[4 + e] += __uint_as_float(ra[it][1][e]); }
            } else {
              float f[8];
              unpack8x<DT, PL>(ra[it][0], ra[it][1], f);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += f[e];
            }
          }
          if (r2) {
            if (r2f) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { v[e] += __uint_as_float(rb[it][0][e]); v[4 + e] += __uint_as_float(rb[it][1][e]); }
            } else {
              float f[8];
              unpack8x<DT, PL>(rb[it][0], rb[it][1], f);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += f[e];
            }
          }
          if (p.row_stats != nullptr) {  // LayerNorm fold, producer side: (sum, sum of squares) per 128-column block
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { sm += v[e]; sq = fmaf(v[e], v[e], sq); }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { sm += __shfl_xor(sm, o, 64); sq += __shfl_xor(sq, o, 64); }  // fixed order
            if (ok[it] && (cn & 15) == 0)
              ((float2*)p.row_stats)[(long long)crow_[it] * p.stats_nblk + ((n0 + cn * 8) >> 7)] = make_float2(sm, sq);
          }
          if (ok[it]) {
            if (p.C16 != nullptr) store8f<DT, 1>((uint16_t*)p.C16 + coff[it], 0, v);
            if (p.c_fp32) {
              float* cp = (float*)p.C + coff[it];
              *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
              *(float4*)(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else if (PL == 2 && p.c_hi_only) {
              store8f<DT, 1>((uint16_t*)p.C + coff[it], 0, v);
            } else {
              store8f<DT, PL>((uint16_t*)p.C + coff[it], p.planes.act, v);
            }
            if (p.C8 != nullptr) {  // e4m3 copy for an fp8 consumer (ReLU'd first when every consumer pre-activates)
              const float qs = p.q_scale != 0.f ? p.q_scale : 1.0f;
              if (p.q_relu) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
              }
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] *= qs;
              *(uint2*)((uint8_t*)p.C8 + coff[it]) = pack_fp8x8(v);
            }
          }
        }
      }
      if (s == 0) DPTX_ESTAMP(2);
    }
    DPTX_ESTAMP(3);
  };
  using std::integral_constant;
  typedef integral_constant<int, 0> I0;
  typedef integral_constant<int, 1> I1;
  typedef integral_constant<int, 2> I2;
  typedef integral_constant<int, -1> IX;
  const bool has1 = p.R1 != nullptr, has2 = p.R2 != nullptr;
  if (p.ln_stats != nullptr) body(I0{}, I0{}, I0{}, I1{});  // qkv, fc1 with the LayerNorm folded in (launch_gemm checks: no residuals)
  else if (!has1 && !has2 && !p.bias_per_img) body(I0{}, I0{}, I0{}, I0{});  // qkv, fc1, most convs
  else if (has1 && p.r1_fp32 && !has2 && !p.bias_per_img) body(I2{}, I0{}, I0{}, I0{});  // proj, fc2: fp32 stream
  else if (has1 && !p.r1_fp32 && !has2 && !p.bias_per_img) body(I1{}, I0{}, I0{}, I0{});  // RCU conv2
  else if (has1 && !p.r1_fp32 && has2 && !p.r2_fp32 && !p.bias_per_img) body(I1{}, I1{}, I0{}, I0{});  // RCU conv2 + path
  else body(IX{}, IX{}, IX{}, I0{});  // patch-embed, readout
#undef DPTX_ESTAMP
}

// ------------------------------------------------------------------- direct-to-LDS kernel
constexpr unsigned OOB = 0x80000000u;  // >= any buffer size we bind (a_bytes < 2^31): reads as zero

// PLE: planes of the EPILOGUE (residual reads, C stores).  PLE == PL by default; PLE == 2 on a single-plane (PL == 1)
// kernel is the per-layer precision policy's "single-pass compute, two-plane tensors": a layer that multiplies hi planes
// only may still read the lo planes of its residuals and write a lo plane for a 3-MFMA consumer
// (GemmParams::c_hi_only / r1_hi_only / r2_hi_only switch the individual planes off).
template <int DT, int BM, int BN, int WAVES_M, int WAVES_N, bool RELU_A, int PL, int PLE = PL, int XT = 3>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, PL == 2 ? 1 : 2) void gemm_glds_kernel(const GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the buffer/LDS-DMA builtins exist only in the device pass
  constexpr int NT = 64 * WAVES_M * WAVES_N;  // 256 threads (2 blocks/CU), or 512 for the 256x256 tile (1 block/CU)
  static_assert(NT == 256 || NT == 512, "4 or 8 waves per block");
  constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
  constexpr int ROWS_PP = NT / 8;                // tile rows one loader pass covers (8 threads x 16 B per 128-B row)
  constexpr int PASS_BYTES = ROWS_PP * 128;
  constexpr int A_PASSES = BM / ROWS_PP, B_PASSES = BN / ROWS_PP;
  constexpr int HK = (TM * TN > 4) ? 2 : BK / 16;
  constexpr int ES = DT == DT_FP8 ? 1 : 2;   // bytes per A / W element
  constexpr int KE = 128 / ES;               // elements of K per k-tile (a tile row is 128 bytes)
  constexpr int DTS = DT == DT_FP8 ? DT_BF16 : DT;  // type of C / residuals
  constexpr int SLABS = (BM * (BN + 4) * 4 > 160 * 1024) ? TM : 1;
  // stage image: [A hi][A lo (PL==2)][W hi][W lo (PL==2)]
  constexpr int A_LO = BM * 128, B_BASE = PL * BM * 128, B_LO = BN * 128;
  constexpr int STAGE_BYTES = PL * (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lr = lane & 31, lh = lane >> 5;

  // block -> tile: XCD x = blockIdx % 8 (observed dispatch placement; only speed depends on it) owns the
  // (x / xcd_n)-th m-slice and (x % xcd_n)-th n-slice of the tile grid; inside a slice n runs fastest so
  // consecutive blocks of one XCD share their A rows.
  const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
  int m0, n0;
  {
    const int x = (int)blockIdx.x & 7, l = (int)blockIdx.x >> 3;
    const int tn_per = tiles_n / p.xcd_n, tm_per = (tiles_m + p.xcd_m - 1) / p.xcd_m;
    const int mt = (x / p.xcd_n) * tm_per + l / tn_per;
    const int nt = (x % p.xcd_n) * tn_per + l % tn_per;
    if (mt >= tiles_m || l >= tm_per * tn_per) return;
    m0 = mt * BM;
    n0 = nt * BN;
  }

  // loader: thread (r0 = tid>>3, kc = tid&7) owns LDS chunk kc of rows r0 + 32*i and fetches the
  // SOURCE chunk kc ^ ((r0>>1)&7)   ((row>>1)&7 is the same for every pass: 32*i leaves bits 1..3)
  const int kc = tid & 7, r0 = tid >> 3;
  const int sc = kc ^ ((r0 >> 1) & 7);
  int a_iy0[A_PASSES], a_ix0[A_PASSES];
  unsigned a_off[A_PASSES];  // byte offset of (img, iy0, ix0, source chunk), mod 2^32
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int m = m0 + r0 + ROWS_PP * i;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    int rem, ox;  // M < 2^23 on this path (launch_cfg): reciprocal division
    const int img = row_div(mm, p.a_rpi, p.a_rpi_rcp, false, rem);
    const int oy = row_div(rem, p.Wout, p.wout_rcp, false, ox);
    a_iy0[i] = ok ? oy * p.stride - p.pad_t : -0x40000000;  // rows >= M never pass the bounds test
    a_ix0[i] = ox * p.stride - p.pad_l;
    const long long e = (long long)img * p.a_img_stride + p.a_off +
                        ((long long)a_iy0[i] * p.Win + a_ix0[i]) * p.a_pix_stride + sc * (16 / ES);
    a_off[i] = (unsigned)(ok ? e * ES : 0);
  }
  unsigned w_off[B_PASSES];
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) w_off[j] = (unsigned)(((long long)(n0 + r0 + ROWS_PP * j) * p.ldw + sc * (16 / ES)) * ES);

  const int w_bytes = (int)((long long)p.N * p.ldw * ES);
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, w_bytes, 0x00020000);
  // lo planes (bf16x3 mode): same offsets, bases shifted by the plane distance
  const __amdgpu_buffer_rsrc_t rsrcAl = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((uint16_t*)const_cast<void*>(p.A) + (PL == 2 ? p.planes.act : 0)), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcWl = __builtin_amdgcn_make_buffer_rsrc(
      (void*)((uint16_t*)const_cast<void*>(p.W) + (PL == 2 ? p.planes.w : 0)), 0, w_bytes, 0x00020000);

  int ky = 0, kx = 0, c0 = 0;  // tap / channel offset of the k tile being LOADED (wave-uniform)

#define DPTX_ISSUE_TILE(BUF, K0)                                                                                   \
  do {                                                                                                             \
    char* sa_ = smem + (BUF) * STAGE_BYTES + wave * 1024;                                                          \
    char* sb_ = sa_ + B_BASE;                                                                                      \
    const unsigned tap_ = (unsigned)(((ky * p.Win + kx) * p.a_pix_stride + c0) * ES);                              \
    const unsigned wk_ = (unsigned)((((ky * p.ksz + kx) * p.Cin) + c0) * ES);  /* k = (tap, channel) in W */          \
    _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                                         \
      const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;                                                            \
      const bool valid = ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);                     \
      const unsigned vo = valid ? a_off[i] + tap_ : OOB;                                                           \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (__attribute__((address_space(3))) void*)(sa_ + i * PASS_BYTES), 16, vo, 0, \
                                               0, 0);                                                              \
      if (PL == 2 && XT == 3)                                                                                      \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcAl, (__attribute__((address_space(3))) void*)(sa_ + A_LO + i * PASS_BYTES), \
                                                 16, vo, 0, 0, 0);                                                 \
    }                                                                                                              \
    _Pragma("unroll") for (int j = 0; j < B_PASSES; ++j) {                                                         \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void*)(sb_ + j * PASS_BYTES), 16,   \
                                               w_off[j] + wk_, 0, 0, 0);                                           \
      if (PL == 2)                                                                                                 \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcWl, (__attribute__((address_space(3))) void*)(sb_ + B_LO + j * PASS_BYTES), \
                                                 16, w_off[j] + wk_, 0, 0, 0);                                     \
    }                                                                                                              \
    if (p.k_tap_fast) { /* the nine taps re-read the same input lines in nine consecutive k-tiles (L2-resident) */ \
      if (++kx == p.ksz) {                                                                                         \
        kx = 0;                                                                                                    \
        if (++ky == p.ksz) { ky = 0; c0 += KE; }                                                                   \
      }                                                                                                            \
    } else { /* channels-fastest inside a tap */                                                                   \
      c0 += KE;                                                                                                    \
      if (c0 >= p.Cin) {                                                                                           \
        c0 = 0;                                                                                                    \
        if (++kx == p.ksz) { kx = 0; ++ky; }                                                                       \
      }                                                                                                            \
    }                                                                                                              \
  } while (0)

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / KE;
  // debug trace (p.trace): per k-tile, lane 0 of every wave of block 0 stamps s_memtime after the barrier, after the DMA
  // issue, after the MFMAs and after the wait for the next tile: where does an iteration's time go?
  const bool tr = p.trace != nullptr && blockIdx.x == 0 && lane == 0;
  long long* trp = p.trace + wave * 4 * 64;
#ifdef DPTX_TRACE   // experiment builds only (build.py DPTX_CXXFLAGS=-DDPTX_TRACE): the stamps cost registers and issue slots
#define DPTX_STAMP(SLOT) do { if (tr && kt < 64) trp[kt * 4 + (SLOT)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define DPTX_STAMP(SLOT) do { } while (0)
#endif
#ifdef DPTX_TRACE
  if (tr && wave == 0) { trp[62 * 4 + 0] = (long long)__builtin_readcyclecounter(); trp[62 * 4 + 1] = (long long)wall_clock64(); }
#endif
  DPTX_ISSUE_TILE(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed (every wave waits for its own DMA, then the barrier publishes all of
    // them) and every wave is done reading the other stage, which the next DMA overwrites
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DPTX_STAMP(3);
    __syncthreads();
    DPTX_STAMP(0);
    const char* sa = smem + (kt & 1) * STAGE_BYTES;
    if (kt + 1 < nk) DPTX_ISSUE_TILE((kt + 1) & 1, (kt + 1) * BK);
    DPTX_STAMP(1);
    mma_tile<DT, TM, TN, RELU_A, PL, HK, XT>(sa, sa + B_BASE, A_LO, B_LO, wm, wn, lr, lh, acc);
    DPTX_STAMP(2);
  }
#ifdef DPTX_TRACE
  if (tr && wave == 0) { trp[62 * 4 + 2] = (long long)__builtin_readcyclecounter(); trp[62 * 4 + 3] = (long long)wall_clock64(); }
#endif
#undef DPTX_STAMP
#undef DPTX_ISSUE_TILE
  __syncthreads();  // all waves finished reading the stages: re-use LDS for the C tile
  epilogue<DTS, BM, BN, TM, TN, PLE, NT, SLABS>(p, smem, m0, n0, wm, wn, lr, lh, tid, acc);
#endif
}

// Quarter-step software pipeline of the fragment reads for the 128x64 wave tile (TM = 4, TN = 2) of the 256x256 kernel:
// the six ds_read_b128 of k-step q+1 are in flight under the eight MFMAs of k-step q (two fragment sets = 48 registers,
// what mma_tile's HK = 2 grouping holds as well), so only the first read of a k-tile is exposed.
struct PpFrags { u32x4_t a[4], b[2]; };
// sa: the wave group's 128-row A tile, sb: the 256-row W tile
__device__ __forceinline__ void pp_read(PpFrags& f, const char* sa, const char* sb, int wn, int lr, int lh, int ks) {
  const int chunk = 2 * ks + lh;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 32 + lr;
    f.a[i] = *(const u32x4_t*)(sa + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int row = wn * 64 + j * 32 + lr;
    f.b[j] = *(const u32x4_t*)(sb + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
  }
}
// ZC: the tile's very first k-step -- the MFMAs take the constant 0 as their C operand, so the 128 accumulator registers
// need no zeroing between tiles (128 v_mov per wave and tile in the persistent loop)
// SWAP: the MFMA takes the W fragment as its A operand and the A fragment as its B operand, i.e. it produces the block
// transposed -- a lane then holds ONE row of C (lane % 32) and, per block, 16 columns in four groups of four consecutive
// ones (column 8 g + 4 (lane / 32) + e in register 4 g + e): what epilogue_direct writes out without the LDS round trip.
template <int DT, bool RELU_A, bool ZC = false, bool SWAP = false>
__device__ __forceinline__ void pp_mma(PpFrags& f, f32x16_t (&acc)[4][2]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (RELU_A) f.a[i] = relu8(f.a[i]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (ZC) {
        f32x16_t z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        acc[i][j] = SWAP ? T16<DT>::mfma32(f.b[j], f.a[i], z) : T16<DT>::mfma32(f.a[i], f.b[j], z);
      } else {
        acc[i][j] = SWAP ? T16<DT>::mfma32(f.b[j], f.a[i], acc[i][j]) : T16<DT>::mfma32(f.a[i], f.b[j], acc[i][j]);
      }
    }
  }
}
// k-steps 0..3 of one k-tile; f0 already holds (or has in flight) k-step 0.  ZC: the first k-tile of an output tile
template <int DT, bool RELU_A, bool ZC = false, bool SWAP = false>
__device__ __forceinline__ void pp_mma_tile(PpFrags& f0, PpFrags& f1, const char* sa, const char* sb, int wn, int lr, int lh,
                                            f32x16_t (&acc)[4][2]) {
  pp_read(f1, sa, sb, wn, lr, lh, 1);
  __builtin_amdgcn_sched_barrier(0);
  pp_mma<DT, RELU_A, ZC, SWAP>(f0, acc);
  __builtin_amdgcn_sched_barrier(0);
  pp_read(f0, sa, sb, wn, lr, lh, 2);
  __builtin_amdgcn_sched_barrier(0);
  pp_mma<DT, RELU_A, false, SWAP>(f1, acc);
  __builtin_amdgcn_sched_barrier(0);
  pp_read(f1, sa, sb, wn, lr, lh, 3);
  __builtin_amdgcn_sched_barrier(0);
  pp_mma<DT, RELU_A, false, SWAP>(f0, acc);
  __builtin_amdgcn_sched_barrier(0);
  pp_mma<DT, RELU_A, false, SWAP>(f1, acc);
}

// Epilogue of the transposed-accumulator form of gemm_pp_kernel (DIRECT): bias / LayerNorm fold / ReLU / GELU and a
// 16-bit store straight from the registers.  A lane owns row lane % 32 of each of its four 32-row blocks; the two lanes
// lane and lane ^ 32 hold interleaved groups of four columns, and one v_permlane32_swap per register pair hands each of
// them eight CONSECUTIVE columns -- a 16-byte store, 32 bytes contiguous per row and instruction.  No accumulator goes
// through LDS (the staged epilogue moves 512 KB per tile through it and takes 12-18 k cycles of a 57-72 k-cycle K = 768
// tile); LDS only holds the tile's 256 bias / column-sum values and the (mu, rstd) row table.
// Launches it serves (launch_cfg): no row remap, no per-image bias, no GroupNorm statistics, 16-bit C, one plane.
// R1 (round 6: proj / fc2 on the 16-bit token stream, the second convolution of a RCU): ONE 16-bit residual with C's rows and
// columns -- it may BE C (the in-place stream: a lane reads exactly the 16 bytes it later writes).  All 16 loads of a lane's
// tile share are issued before anything else (64 registers: the k-loop's fragment registers are free here), so their latency
// runs under the bias fill, the wait for the next tile's DMA and the barrier -- once per tile.  With R1 the launch may also
// be the PRODUCER side of the LayerNorm fold (p.row_stats: (sum, sum of squares) of every row per 128-column block): a lane
// sums its eight columns in order, then the staged epilogue's butterfly over the 16 eight-column chunks of a block is
// replayed level by level -- chunk c = 8 (wn & 1) + 4 j + 2 pr + lh, so level 1 is the lane ^ 32 partner, levels 2 and 3 are
// in-lane (pr, j) and level 4 pairs the waves wn, wn ^ 1 through LDS: the same additions in the same association, the
// records are bit-identical to the staged form's (tests/test_gpu_ops.py::test_gemm_launch_forms_agree_bit_for_bit).
// RES: 0 no residual; 1 R1 16-bit (all loads up front); 2 R1 + R2 16-bit (RCU conv2 + the fusion path: both one 32-row block
// ahead -- all of them up front would be 128 registers next to the 128 accumulators).  (The fp32 token stream of the parity
// mode -- R1 and C fp32, plus a 16-bit copy -- was built in this form too and is SLOWER than the staged epilogue: 10 bytes
// per element in 32-byte pieces per row and instruction; profiles/r06_experiments.md section 8.  It stays staged.)
template <int DT, int RES>
__device__ __forceinline__ void epilogue_direct(const GemmParams& p, char* smem, int m0, int n0, int wm, int wn, int lr, int lh,
                                                int tid, f32x16_t (&acc)[4][2], bool dma_in_flight, const char* lds_ln) {
  constexpr bool R1 = RES != 0;
  float* sbias = (float*)smem;          // [256]
  float* scol = sbias + 256;            // [256] column sums of the folded weight
  float2* lnrow = (float2*)(scol + 256);  // [256] (mu, rstd)
  float2* spart = lnrow + 256;          // [256 rows][4 wave columns] row-statistics partials (R1 && row_stats)
  const bool lnf = p.ln_stats != nullptr;
  u32x4_t rres[4][2][2];      // RES 1: R1 of the lane's whole tile share
  u32x4_t rnext[2][2][2][2];  // RES 2: block i in set i & 1, [set][j][pr][R1 / R2]
  auto row_off = [&](int i) -> long long {
    const int m = m0 + wm * 128 + i * 32 + lr;
    return ((long long)p.c_row_off + (m < p.M ? m : m0)) * p.ldc + n0 + wn * 64 + 8 * lh;
  };
  auto load_next = [&](int i) {   // i: compile-time after unrolling
    const long long ro = row_off(i);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const long long o = ro + j * 32 + 16 * pr;
        if constexpr (RES == 2) {
          rnext[i & 1][j][pr][0] = *(const u32x4_t*)((const uint16_t*)p.R1 + o);
          rnext[i & 1][j][pr][1] = *(const u32x4_t*)((const uint16_t*)p.R2 + o);
        }
      }
  };
  if constexpr (RES == 1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint16_t* rrow = (const uint16_t*)p.R1 + row_off(i);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) rres[i][j][pr] = *(const u32x4_t*)(rrow + j * 32 + 16 * pr);
    }
  }
  if constexpr (RES == 2) load_next(0);
  if (tid < 256) {
    sbias[tid] = p.bias != nullptr ? p.bias[n0 + tid] : 0.f;
    scol[tid] = lnf ? p.ln_colsum[n0 + tid] : 0.f;
  } else if (lnf) {
    const int r = tid - 256;
    const bool all8 = p.ln_nblk == 8;
    int m = m0 + r;
    m = m < p.M ? m : m0;
    const float4* st = lds_ln != nullptr ? (const float4*)(lds_ln + r * 64) : (const float4*)(p.ln_stats + (long long)m * 16);
    const float4 r0 = st[0], r1_ = st[1], r2_ = st[2], r3_ = st[3];
    // same association as the staged epilogue's table: the two forms give bit-identical results
    const float sm = (r0.x + r0.z) + (r1_.x + r1_.z) + ((r2_.x + r2_.z) + (all8 ? r3_.x + r3_.z : 0.f));
    const float sq = (r0.y + r0.w) + (r1_.y + r1_.w) + ((r2_.y + r2_.w) + (all8 ? r3_.y + r3_.w : 0.f));
    lnrow[r] = ln_mu_rstd(sm, sq, p.ln_inv_dim, p.ln_eps);
  }
  // the next tile's first k-tile (DMA issued by the caller) lands before this tile's first store is issued: stores count
  // in vmcnt, a vmcnt(0) after them would wait for the tile to reach memory
  if (dma_in_flight) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const bool stats = R1 && p.row_stats != nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wm * 128 + i * 32 + lr;
    const int m = m0 + row;
    const bool ok = m < p.M;
    float csm[2][2], csq[2][2];
    if constexpr (RES == 2) {
      if (i + 1 < 4) load_next(i + 1);
    }
    float2 ms = make_float2(0.f, 1.f);
    if (lnf) ms = lnrow[row];
    uint16_t* crow = (uint16_t*)p.C + ((long long)p.c_row_off + (ok ? m : m0)) * p.ldc + n0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // lanes 0..31 keep their even group and receive the partner's; lanes 32..63 likewise with the odd group
          const u32x2_t sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][j][8 * pr + e]),
                                                              __float_as_uint(acc[i][j][8 * pr + 4 + e]), false, false);
          v[e] = __uint_as_float(sw.x);
          v[4 + e] = __uint_as_float(sw.y);
        }
        const int col = wn * 64 + j * 32 + 8 * (2 * pr + lh);
        const float4 b0 = *(const float4*)(sbias + col), b1 = *(const float4*)(sbias + col + 4);
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        if (lnf) {  // the same two explicit fused multiply-adds as the staged epilogue
          const float4 c0 = *(const float4*)(scol + col), c1 = *(const float4*)(scol + col + 4);
          const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = ln_fold_fma(v[e], ms.x, ms.y, cc[e], bb[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += bb[e];
        }
        if (p.act == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        } else if (p.act == 2) {
#pragma unroll
          for (int e = 0; e < 8; e += 2) gelu_erf2(v[e], v[e + 1]);
        }
        if constexpr (R1) {
          {
            float f[8];
            if constexpr (RES == 2) unpack8x<DT, 1>(rnext[i & 1][j][pr][0], rnext[i & 1][j][pr][0], f);
            else unpack8x<DT, 1>(rres[i][j][pr], rres[i][j][pr], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += f[e];
          }
          if constexpr (RES == 2) {
            float f[8];
            unpack8x<DT, 1>(rnext[i & 1][j][pr][1], rnext[i & 1][j][pr][1], f);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += f[e];
          }
          if (stats) {   // chunk sums in column order, as the staged epilogue's threads form them
            float sm = 0.f, sq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { sm += v[e]; sq = fmaf(v[e], v[e], sq); }
            csm[j][pr] = sm;
            csq[j][pr] = sq;
          }
        }
        if (ok) store8f<DT, 1>(crow + col, 0, v);
      }
    }
    if constexpr (R1) {
      if (stats) {
        float sm[2], sq[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float a[2], b[2];
#pragma unroll
          for (int pr = 0; pr < 2; ++pr) {   // level 1: chunk c ^ 1 is the lane ^ 32 partner's
            a[pr] = csm[j][pr] + __shfl_xor(csm[j][pr], 32, 64);
            b[pr] = csq[j][pr] + __shfl_xor(csq[j][pr], 32, 64);
          }
          sm[j] = a[0] + a[1];               // level 2: pr
          sq[j] = b[0] + b[1];
        }
        if (lh == 0) spart[row * 4 + wn] = make_float2(sm[0] + sm[1], sq[0] + sq[1]);   // level 3: j
      }
    }
  }
  if constexpr (R1) {
    if (stats) {
      __syncthreads();
      // level 4: the two waves of a 128-column block; wave (wm, wn even) writes the records of its rows' block wn / 2
      if ((wn & 1) == 0 && lh == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = wm * 128 + i * 32 + lr;
          const int m = m0 + row;
          const float2 x = spart[row * 4 + wn], y = spart[row * 4 + wn + 1];
          if (m < p.M)
            ((float2*)p.row_stats)[((long long)p.c_row_off + m) * p.stats_nblk + ((n0 + wn * 64) >> 7)] = make_float2(x.x + y.x, x.y + y.y);
        }
      }
    }
  }
}

// ------------------------------------------------------------------- ping-pong 256x256 kernel
// The k-loop trace of gemm_glds_kernel (profiles/r02_gemm_trace.txt) shows what bounds a lockstep 256x256 tile: an
// iteration is 4100 cycles for 2048 cycles of MFMA work per SIMD, because the eight waves all issue their 8 LDS-DMA
// instructions first (~100-170 cycles EACH, during which a wave issues nothing else), then all read fragments, then the
// two waves of every SIMD queue their MFMAs behind each other.
//
// Here the two wave groups (wm = 0 / 1: the upper / lower 128 rows of the tile, one wave of each on every SIMD) run the
// SAME work half an iteration apart, two barriers per k-tile:
//     slot 1:  group 0 issues DMA for tile t+1          |  group 1 multiplies tile t
//     slot 2:  group 0 multiplies tile t                |  group 1 issues DMA for tile t+1
// so a SIMD always has one wave feeding the matrix pipe while the other is stuck in the DMA issue.  Who loads what follows
// from who needs it first: group 1 multiplies tile t+1 in the very next slot after group 1's own issue slot, so everything
// group 1 reads -- A rows 128..255 and all of W -- is issued by GROUP 0 one full slot earlier (12 instructions per wave),
// and group 1 issues only A rows 0..127 (4 instructions), which group 0 reads a full iteration later.  Every wave drains
// its own DMA (vmcnt(0)) at the end of its MFMA slot, i.e. before the barrier in front of the first reader.
// (Round 2 measured seven variants of this schedule -- third LDS buffer, all DMA on one group, one barrier per k-tile with
// overlapping MFMA slots, s_setprio, the guide's 8-phase structure, a 256x128 three-stage tile, an LDS-resident conv halo
// -- all <= 0: profiles/r02_experiments.md; the three that were kernels of their own are in the git history, round 5's tree.)
// DIRECT: transposed accumulators + epilogue_direct (the launches with a plain epilogue: qkv, fc1, the first convolution of a
// RCU, layerN_rn, output_conv.0); everything else about the kernel is the same.
template <int DT, bool RELU_A, int PLE = 1, int DIRECT = 0>   // DIRECT: 0 staged epilogue, 1 + RES: epilogue_direct<DT, RES>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 256, BN = 256, NT = 512, TM = 4, TN = 2, PL = 1;
  constexpr int SLABS = TM;
  constexpr int HALF = 128 * 128;  // bytes of a 128-row operand tile
  // LDS: [stage 0 | stage 1, re-used by the epilogue: 8 x 32 x 72 floats + the (mu, rstd) row table | statistics records]
  constexpr int LN_LDS = 4 * HALF + 8 * 32 * 72 * 4 + BM * 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // Thread coordinates.  They are re-derived from an opaque copy of threadIdx at the top of every tile (DPTX_PP_COORDS) so
  // that none of them -- nor anything computed from them -- has to stay in a register across the epilogue, whose slab loop
  // needs every register next to the 128 accumulators.
  int tid = threadIdx.x;
  int lane, wave, wm, wn, lr, lh;   // wm = wave group
  int t, kc, r0, sc, a_row0, wq;    // loader coordinates, below
#define DPTX_PP_COORDS()                                                                                           \
  do {                                                                                                             \
    asm volatile("" : "+v"(tid));                                                                                  \
    lane = tid & 63;                                                                                               \
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);                                                               \
    wm = wave >> 2; wn = wave & 3;                                                                                 \
    lr = lane & 31; lh = lane >> 5;                                                                                \
    t = tid & 255;                                                                                                 \
    kc = t & 7; r0 = t >> 3;                                                                                       \
    sc = kc ^ ((r0 >> 1) & 7);                                                                                     \
    a_row0 = wm == 0 ? 128 : 0;                                                                                    \
    wq = wave & 3;     /* wave inside its group: rows 8*wq .. 8*wq+7 of every 32-row pass */                       \
  } while (0)
  DPTX_PP_COORDS();
  // LDS map: stage b = [A rows 0..127 | A rows 128..255 | W rows 0..255] at b * 64 KB
  auto w_ptr = [&](int b) -> char* { return smem + b * 4 * HALF + 2 * HALF; };
  auto alo_ptr = [&](int b) -> char* { return smem + b * 4 * HALF; };
  auto ahi_ptr = [&](int b) -> char* { return smem + b * 4 * HALF + HALF; };

  // Tile loop: XCD x = blockIdx % 8 owns a 2-D slice of the tile grid (launch_cfg); block l of the XCD takes the slice's
  // tiles l, l + L, l + 2L, ... (L = gridDim / 8 blocks per XCD).  A launch with one block per tile (L = tiles per slice)
  // runs the loop once; the persistent launch (L = 32: one block per CU) keeps the block and overlaps a tile's epilogue --
  // the LDS transposition of the accumulators, the residual loads, the stores draining to HBM -- with the landing of the
  // next tile's first k-tile.  Which block computes a tile does not change the tile's arithmetic: results are bit-identical.
  const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
  const int xcd = (int)blockIdx.x & 7, lstep = (int)gridDim.x >> 3;
  const int tn_per = tiles_n / p.xcd_n, tm_per = (tiles_m + p.xcd_m - 1) / p.xcd_m;
  const int ltot = tm_per * tn_per;
  const int mt_base = (xcd / p.xcd_n) * tm_per, nt_base = (xcd % p.xcd_n) * tn_per;
  int l = (int)blockIdx.x >> 3;
  if (l >= ltot || mt_base + l / tn_per >= tiles_m) return;
  int m0, n0;

  // loader of a group: thread (r0 = t>>3, kc = t&7), t = tid % 256, owns LDS chunk kc of rows base + r0 + 32*i.
  // Group 0 loads A rows 128..255 (4 passes) and W rows 0..255 (8 passes); group 1 loads A rows 0..127 (4 passes).
  int a_iy0[4], a_ix0[4];
  unsigned a_off[4];
  unsigned w_off[8];
  // addressing of tile `l` of this XCD's slice
#define DPTX_PP_TILE_SETUP()                                                                                       \
  do {                                                                                                             \
    m0 = (mt_base + l / tn_per) * BM;                                                                              \
    n0 = (nt_base + l % tn_per) * BN;                                                                              \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                \
      const int m = m0 + a_row0 + r0 + 32 * i;                                                                     \
      const bool ok = m < p.M;                                                                                     \
      const int mm = ok ? m : 0;                                                                                   \
      int rem, ox;                                                                                                 \
      const int img = row_div(mm, p.a_rpi, p.a_rpi_rcp, false, rem);                                               \
      const int oy = row_div(rem, p.Wout, p.wout_rcp, false, ox);                                                  \
      a_iy0[i] = ok ? oy * p.stride - p.pad_t : -0x40000000;                                                       \
      a_ix0[i] = ox * p.stride - p.pad_l;                                                                          \
      /* element offset mod 2^32 (the buffer is < 2^31 bytes: launch_cfg) */                                       \
      const unsigned e = (unsigned)img * (unsigned)p.a_img_stride + (unsigned)p.a_off +                            \
                         (unsigned)(a_iy0[i] * p.Win + a_ix0[i]) * (unsigned)p.a_pix_stride + (unsigned)(sc * 8);  \
      a_off[i] = ok ? e * 2u : 0u;                                                                                 \
    }                                                                                                              \
    /* byte offsets mod 2^32 (W is < 2^31 bytes: launch_cfg) */                                                    \
    _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                                  \
      w_off[j] = ((unsigned)(n0 + r0 + 32 * j) * (unsigned)p.ldw + (unsigned)(sc * 8)) * 2u;                       \
  } while (0)
  const int w_bytes = (int)((long long)p.N * p.ldw * 2);
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, w_bytes, 0x00020000);

  // tap / channel offset (wave-uniform) of the k-tile whose A rows this group loads next, and (group 0) whose W rows
  int ky = 0, kx = 0, c0 = 0, kyw = 0, kxw = 0, c0w = 0;
#define DPTX_PP_NEXT(KY, KX, C0)                                                                                   \
  do {                                                                                                             \
    if (p.k_tap_fast) {                                                                                            \
      if (++KX == p.ksz) { KX = 0; if (++KY == p.ksz) { KY = 0; C0 += BK; } }                                      \
    } else {                                                                                                       \
      C0 += BK;                                                                                                    \
      if (C0 >= p.Cin) { C0 = 0; if (++KX == p.ksz) { KX = 0; ++KY; } }                                            \
    }                                                                                                              \
  } while (0)
  // this group's four A pieces of the tile at (ky, kx, c0) into the 128-row tile at DST; advances the tap
#define DPTX_PP_ISSUE_A(DST)                                                                                       \
  do {                                                                                                             \
    char* d_ = (DST) + wq * 1024;                                                                                  \
    const unsigned tap_ = (unsigned)(((ky * p.Win + kx) * p.a_pix_stride + c0) * 2);                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                \
      const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;                                                            \
      const bool valid = ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);                     \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (__attribute__((address_space(3))) void*)(d_ + 32 * i * 128), 16, \
                                               valid ? a_off[i] + tap_ : OOB, 0, 0, 0);                            \
    }                                                                                                              \
    DPTX_PP_NEXT(ky, kx, c0);                                                                                      \
  } while (0)
  // the eight W pieces of the tile at (kyw, kxw, c0w) into the 256-row tile at DST; advances that tap
#define DPTX_PP_ISSUE_W(DST)                                                                                       \
  do {                                                                                                             \
    char* d_ = (DST) + wq * 1024;                                                                                  \
    const unsigned wk_ = (unsigned)((((kyw * p.ksz + kxw) * p.Cin) + c0w) * 2);                                    \
    _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                                  \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void*)(d_ + 32 * j * 128), 16, \
                                               w_off[j] + wk_, 0, 0, 0);                                           \
    DPTX_PP_NEXT(kyw, kxw, c0w);                                                                                   \
  } while (0)

  const int nk = p.K / BK;
  const bool tr = p.trace != nullptr && blockIdx.x == 0 && lane == 0;
  long long* trp = p.trace + wave * 4 * 64;
#ifdef DPTX_TRACE   // experiment builds only (build.py DPTX_CXXFLAGS=-DDPTX_TRACE): the stamps cost registers and issue slots
#define DPTX_STAMP(SLOT) do { if (tr && kt < 40) trp[kt * 4 + (SLOT)] = (long long)__builtin_readcyclecounter(); } while (0)
  if (tr && wave == 0) { trp[62 * 4 + 0] = (long long)__builtin_readcyclecounter(); trp[62 * 4 + 1] = (long long)wall_clock64(); }
#else
#define DPTX_STAMP(SLOT) do { } while (0)
  (void)tr; (void)trp;
#endif
  // the first k-tile (tap 0, channel 0) of a tile into stage 0.  Nobody multiplies before all of it has landed, so the
  // sixteen pieces are split evenly -- each group its own A rows and four of the eight W pieces (the k-loop's split, 12 : 4,
  // follows from who reads what first); group 0 keeps the W tap state
#define DPTX_PP_PROLOGUE()                                                                                         \
  do {                                                                                                             \
    ky = kx = c0 = kyw = kxw = c0w = 0;                                                                            \
    {                                                                                                              \
      char* d_ = w_ptr(0) + wq * 1024;                                                                             \
      const int jb_ = wm == 0 ? 0 : 4;                                                                             \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void*)(d_ + 32 * (jb_ + j) * 128), \
                                                 16, wm == 0 ? w_off[j] : w_off[4 + j], 0, 0, 0);                  \
    }                                                                                                              \
    DPTX_PP_NEXT(kyw, kxw, c0w);                                                                                   \
    if (wm == 0) DPTX_PP_ISSUE_A(ahi_ptr(0));                                                                      \
    else DPTX_PP_ISSUE_A(alo_ptr(0));                                                                              \
  } while (0)
#ifdef DPTX_TRACE   // tile phases (thread 0 of block 0): row 40 + tile of wave 0's trace block
  int ti = 0;
#define DPTX_TSTAMP(SLOT) do { if (tr && wave == 0 && ti < 8) trp[(40 + ti) * 4 + (SLOT)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define DPTX_TSTAMP(SLOT) do { } while (0)
#endif
  DPTX_PP_TILE_SETUP();
  DPTX_PP_PROLOGUE();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (;;) {
    // the tile's addressing is (re)computed here from l: it must not stay live across the epilogue, whose slab loop needs
    // every register next to the 128 accumulators (the asm makes l opaque, so the values above are not carried over)
    asm volatile("" : "+s"(l));
    DPTX_PP_COORDS();
    DPTX_PP_TILE_SETUP();
    f32x16_t acc[TM][TN];  // not zeroed: the first k-step's MFMAs take C = 0 (the peeled kt = 0 below)
    // every wave has waited for its own pieces of k-tile 0 (above / inside the previous tile's epilogue) and is done with the
    // epilogue's LDS tile, which the DMA of k-tile 1 overwrites
    __syncthreads();
    DPTX_TSTAMP(0);
    // LayerNorm fold, consumer side: the tile's 256 statistics records (64 bytes per row) travel to LDS under the k-loop --
    // two 16-byte pieces per thread, lane-linear = row-major; read by the epilogue's row table instead of a global load whose
    // latency nothing covered
    if (p.ln_stats != nullptr) {
      const __amdgpu_buffer_rsrc_t rsrcS =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.ln_stats), 0, (int)((long long)p.M * 64), 0x00020000);
      _Pragma("unroll") for (int k = 0; k < 2; ++k)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcS, (__attribute__((address_space(3))) void*)(smem + LN_LDS + (wave * 2 + k) * 1024), 16,
                                                 (unsigned)m0 * 64u + (unsigned)((wave * 2 + k) * 1024 + lane * 16), 0, 0, 0);
    }
    // two straight-line loops, one per group (an MFMA under a per-slot branch makes the 128 accumulator registers a phi
    // that hipcc resolves with copies: 500 spilled registers)
    PpFrags f0, f1;
    // one k-tile of group 0 / group 1 (ZC: kt == 0, peeled in front of each loop)
#define DPTX_PP_G0_ITER(ZC)                                                                                        \
  do {                                                                                                             \
    DPTX_STAMP(0);                                                                                                 \
    /* slot 1: the DMA -- W and A rows 128..255 of tile kt+1 */                                                    \
    const char* sa = alo_ptr(kt & 1);                                                                              \
    const char* sb = w_ptr(kt & 1);                                                                                \
    if (kt + 1 < nk) {                                                                                             \
      DPTX_PP_ISSUE_W(w_ptr((kt + 1) & 1));                                                                        \
      DPTX_PP_ISSUE_A(ahi_ptr((kt + 1) & 1));                                                                      \
    }                                                                                                              \
    DPTX_STAMP(1);                                                                                                 \
    asm volatile("s_barrier" ::: "memory");                                                                        \
    DPTX_STAMP(2);                                                                                                 \
    pp_read(f0, sa, sb, wn, lr, lh, 0);            /* slot 2 */                                                    \
    pp_mma_tile<DT, RELU_A, ZC, DIRECT != 0>(f0, f1, sa, sb, wn, lr, lh, acc);                                          \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  /* what group 1 reads in its next slot has landed */         \
    DPTX_STAMP(3);                                                                                                 \
    asm volatile("s_barrier" ::: "memory");                                                                        \
  } while (0)
#define DPTX_PP_G1_ITER(ZC)                                                                                        \
  do {                                                                                                             \
    DPTX_STAMP(0);                                                                                                 \
    const char* sa = ahi_ptr(kt & 1);              /* slot 1 */                                                    \
    const char* sb = w_ptr(kt & 1);                                                                                \
    pp_read(f0, sa, sb, wn, lr, lh, 0);                                                                            \
    pp_mma_tile<DT, RELU_A, ZC, DIRECT != 0>(f0, f1, sa, sb, wn, lr, lh, acc);                                          \
    /* its DMA of the previous slot 2 (A rows 0..127 of THIS tile) has landed before group 0 reads it in slot 2 */ \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                               \
    DPTX_STAMP(1);                                                                                                 \
    asm volatile("s_barrier" ::: "memory");                                                                        \
    DPTX_STAMP(2);                                                                                                 \
    if (kt + 1 < nk) DPTX_PP_ISSUE_A(alo_ptr((kt + 1) & 1));   /* slot 2 */                                        \
    DPTX_STAMP(3);                                                                                                 \
    asm volatile("s_barrier" ::: "memory");                                                                        \
  } while (0)
    if (wm == 0) {
      { const int kt = 0; DPTX_PP_G0_ITER(true); }
      for (int kt = 1; kt < nk; ++kt) DPTX_PP_G0_ITER(false);
    } else {
      { const int kt = 0; DPTX_PP_G1_ITER(true); }
      for (int kt = 1; kt < nk; ++kt) DPTX_PP_G1_ITER(false);
    }
#undef DPTX_PP_G0_ITER
#undef DPTX_PP_G1_ITER
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // every wave is done reading both stages
    DPTX_TSTAMP(1);
    // next tile of this block: its first k-tile flies into stage 0 under this tile's epilogue, which stages the accumulators
    // through the memory of stage 1 (+ 4.6 KB behind it)
    const int em0 = m0, en0 = n0;
    l += lstep;
    const bool more = l < ltot && mt_base + l / tn_per < tiles_m;
    if (more) {
      DPTX_PP_TILE_SETUP();
      DPTX_PP_PROLOGUE();
    }
    DPTX_TSTAMP(2);
    {
      // opaque copies of the thread coordinates: the epilogue's per-thread addresses are the same for every tile, and hoisted
      // out of the tile loop they would occupy ~40 registers through the k-loop (scratch spills)
      int etid = tid, elr = lr, elh = lh, ewm = wm, ewn = wn;
      asm volatile("" : "+v"(etid), "+v"(elr), "+v"(elh), "+s"(ewm), "+s"(ewn));
#ifdef DPTX_TRACE
      const int trow = ti < 8 ? ti : -1;
#else
      const int trow = -1;
#endif
#ifdef DPTX_NO_WP   // A/B builds: block-wide staging everywhere
      constexpr bool wp_ok = false;
#else
      constexpr bool wp_ok = true;
#endif
      if constexpr (DIRECT) {
        (void)trow; (void)wp_ok;
        epilogue_direct<DT, DIRECT - 1>(p, smem + 4 * HALF, em0, en0, ewm, ewn, elr, elh, etid, acc, more,
                            p.ln_stats != nullptr ? smem + LN_LDS : nullptr);
      } else if (wp_ok && p.row_stats == nullptr)
        epilogue<DT, BM, BN, TM, TN, PLE, NT, SLABS, false, true>(p, smem + 4 * HALF, em0, en0, ewm, ewn, elr, elh, etid, acc, 0, more, trow,
                                                                  p.ln_stats != nullptr ? smem + LN_LDS : nullptr);
      else  // the producer side of the LayerNorm fold reduces 128 columns of a row: block-wide staging
        epilogue<DT, BM, BN, TM, TN, PLE, NT, SLABS>(p, smem + 4 * HALF, em0, en0, ewm, ewn, elr, elh, etid, acc, 0, more, trow);
    }
    DPTX_TSTAMP(3);
#ifdef DPTX_TRACE
    ++ti;
#endif
    if (!more) break;
  }
#ifdef DPTX_TRACE
  if (tr && wave == 0) { trp[62 * 4 + 2] = (long long)__builtin_readcyclecounter(); trp[62 * 4 + 3] = (long long)wall_clock64(); }
#endif
#undef DPTX_STAMP
#undef DPTX_TSTAMP
#undef DPTX_PP_PROLOGUE
#undef DPTX_PP_TILE_SETUP
#undef DPTX_PP_COORDS
#undef DPTX_PP_ISSUE_W
#undef DPTX_PP_ISSUE_A
#undef DPTX_PP_NEXT
#endif
}

// ------------------------------------------------------------------- ping-pong 128x128 kernel for hi/lo planes
// Round 4.  In the parity mode the two-plane gemm_glds_kernel<fp16, 128, 128, 2, 4> is 37 % of the forward (59 launches,
// 8.2 ms: profiles/r04_kernel_trace_stats_mixed.txt); its k-loop is the lockstep one -- all eight waves issue their eight DMA
// pieces, then all multiply: 1.7 us per k-tile on output_conv.0 for 1536 cycles of MFMA work per SIMD.  A two-plane stage of
// a 128x128 tile is 512 LDS rows = 64 pieces of 1 KB, exactly what a one-plane stage of the 256x256 tile is, so the schedule
// of gemm_pp_kernel carries over: the two wave groups (wm = 0 / 1: rows 0..63 / 64..127, one wave of each on every SIMD;
// wave tile 64 x 32, 24 MFMAs per k-tile) run half an iteration apart, two barriers per k-tile,
//     slot 1:  group 0 issues W hi / lo and A rows 64..127 hi / lo of tile t+1 (12 pieces per wave)  |  group 1 multiplies tile t
//     slot 2:  group 0 multiplies tile t                         |  group 1 issues A rows 0..63 hi / lo of tile t+1 (4 pieces)
// with the same who-reads-what-first argument.  Stage image: [A0 hi | A0 lo | A1 hi | A1 lo | W hi | W lo] (64 + 64 + 64 + 64 +
// 128 + 128 rows).  One block per tile (no persistent loop), the epilogue is gemm_glds_kernel's.  XT == 2 (a_hi_only): the A lo
// pieces are not issued (10 + 2 per wave) and not read.  DPTX_PP2=0: the lockstep kernel (A/B runs; bit-identical results --
// the MFMAs of an accumulator run in the same order).  Same box, single-stream per-launch events: output_conv.0 2.15 -> 1.92 ms,
// layer2_rn 0.538 -> 0.480, the ResNetV2 3x3 / 1x1 convolutions -8 ... -13 %; 0.63 ms of the 21.8 ms launch sum, parity mode
// 1646 -> 1667 img/s in the two-stream schedule.
template <int DT, bool RELU_A, int XT = 3>
__global__ __launch_bounds__(512, 1) void gemm_pp2_kernel(const GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 128, BN = 128, NT = 512, TM = 2, TN = 1, PL = 2;
  constexpr int Q = 64 * 128;              // bytes of a 64-row operand tile
  constexpr int STAGE = 8 * Q;             // 64 KB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;  // wm = wave group
  const int lr = lane & 31, lh = lane >> 5;
  const int t = tid & 255, kc = t & 7, r0 = t >> 3;   // loader coordinates inside the group: LDS chunk kc of rows r0 + 32 i
  const int sc = kc ^ ((r0 >> 1) & 7);
  const int wq = wave & 3;                  // rows 8 wq .. 8 wq + 7 of every 32-row pass
  auto a_ptr = [&](int b, int g) -> char* { return smem + b * STAGE + g * 2 * Q; };   // [hi 64 rows | lo 64 rows] of group g
  auto w_ptr = [&](int b) -> char* { return smem + b * STAGE + 4 * Q; };              // [hi 128 rows | lo 128 rows]

  const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
  int m0, n0;
  {
    const int x = (int)blockIdx.x & 7, l = (int)blockIdx.x >> 3;
    const int tn_per = tiles_n / p.xcd_n, tm_per = (tiles_m + p.xcd_m - 1) / p.xcd_m;
    const int mt = (x / p.xcd_n) * tm_per + l / tn_per;
    const int nt = (x % p.xcd_n) * tn_per + l % tn_per;
    if (mt >= tiles_m || l >= tm_per * tn_per) return;
    m0 = mt * BM;
    n0 = nt * BN;
  }
  // group 0 loads the A rows of group 1 and vice versa (who reads what first, above)
  const int a_row0 = wm == 0 ? 64 : 0;
  int a_iy0[2], a_ix0[2];
  unsigned a_off[2], w_off[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + a_row0 + r0 + 32 * i;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    int rem, ox;
    const int img = row_div(mm, p.a_rpi, p.a_rpi_rcp, false, rem);
    const int oy = row_div(rem, p.Wout, p.wout_rcp, false, ox);
    a_iy0[i] = ok ? oy * p.stride - p.pad_t : -0x40000000;
    a_ix0[i] = ox * p.stride - p.pad_l;
    const unsigned e = (unsigned)img * (unsigned)p.a_img_stride + (unsigned)p.a_off +
                       (unsigned)(a_iy0[i] * p.Win + a_ix0[i]) * (unsigned)p.a_pix_stride + (unsigned)(sc * 8);
    a_off[i] = ok ? e * 2u : 0u;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) w_off[j] = ((unsigned)(n0 + r0 + 32 * j) * (unsigned)p.ldw + (unsigned)(sc * 8)) * 2u;
  const int w_bytes = (int)((long long)p.N * p.ldw * 2);
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcAl =
      __builtin_amdgcn_make_buffer_rsrc((void*)((uint16_t*)const_cast<void*>(p.A) + p.planes.act), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcWl =
      __builtin_amdgcn_make_buffer_rsrc((void*)((uint16_t*)const_cast<void*>(p.W) + p.planes.w), 0, w_bytes, 0x00020000);

  int ky = 0, kx = 0, c0 = 0, kyw = 0, kxw = 0, c0w = 0;   // tap state of this group's A pieces / of the W pieces
#define DPTX_P2_NEXT(KY, KX, C0)                                                                                   \
  do {                                                                                                             \
    if (p.k_tap_fast) {                                                                                            \
      if (++KX == p.ksz) { KX = 0; if (++KY == p.ksz) { KY = 0; C0 += BK; } }                                      \
    } else {                                                                                                       \
      C0 += BK;                                                                                                    \
      if (C0 >= p.Cin) { C0 = 0; if (++KX == p.ksz) { KX = 0; ++KY; } }                                            \
    }                                                                                                              \
  } while (0)
  // this group's A pieces (the OTHER group's rows) of the k-tile at (ky, kx, c0) into DST = [hi 64 rows | lo 64 rows]
#define DPTX_P2_ISSUE_A(DST)                                                                                       \
  do {                                                                                                             \
    char* d_ = (DST) + wq * 1024;                                                                                  \
    const unsigned tap_ = (unsigned)(((ky * p.Win + kx) * p.a_pix_stride + c0) * 2);                               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                                \
      const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;                                                            \
      const bool valid = ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);                     \
      const unsigned vo = valid ? a_off[i] + tap_ : OOB;                                                           \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (__attribute__((address_space(3))) void*)(d_ + 32 * i * 128), 16, vo, 0, 0, 0); \
      if (XT == 3)                                                                                                 \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcAl, (__attribute__((address_space(3))) void*)(d_ + Q + 32 * i * 128), 16, vo, 0, 0, 0); \
    }                                                                                                              \
    DPTX_P2_NEXT(ky, kx, c0);                                                                                      \
  } while (0)
  // W passes J0 .. J0 + NJ - 1 of [hi: 0..3 | lo: 4..7] of the k-tile at (kyw, kxw, c0w) into the 256-row image at DST
#define DPTX_P2_ISSUE_W(DST, J0, NJ)                                                                               \
  do {                                                                                                             \
    char* d_ = (DST) + wq * 1024;                                                                                  \
    const unsigned wk_ = (unsigned)((((kyw * p.ksz + kxw) * p.Cin) + c0w) * 2);                                    \
    _Pragma("unroll") for (int j = (J0); j < (J0) + (NJ); ++j) {                                                   \
      if (j < 4)                                                                                                   \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void*)(d_ + 32 * j * 128), 16, \
                                                 w_off[j & 3] + wk_, 0, 0, 0);                                     \
      else                                                                                                         \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcWl, (__attribute__((address_space(3))) void*)(d_ + 32 * j * 128), 16, \
                                                 w_off[j & 3] + wk_, 0, 0, 0);                                     \
    }                                                                                                              \
  } while (0)

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
  const int nk = p.K / BK;
  // first k-tile into stage 0: nobody multiplies before all of it has landed, so the pieces are split evenly -- group 0 the
  // hi plane of W, group 1 the lo plane, each group the A rows it loads in the loop as well
  if (wm == 0) DPTX_P2_ISSUE_W(w_ptr(0), 0, 4);
  else DPTX_P2_ISSUE_W(w_ptr(0), 4, 4);
  DPTX_P2_NEXT(kyw, kxw, c0w);
  DPTX_P2_ISSUE_A(a_ptr(0, wm ^ 1));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // two straight-line loops, one per group (an MFMA under a per-slot branch turns the accumulators into a phi)
  if (wm == 0) {
    for (int kt = 0; kt < nk; ++kt) {
      const char* sa = a_ptr(kt & 1, 0);
      const char* sb = w_ptr(kt & 1);
      if (kt + 1 < nk) {                                   // slot 1: the DMA
        DPTX_P2_ISSUE_W(w_ptr((kt + 1) & 1), 0, 8);
        DPTX_P2_NEXT(kyw, kxw, c0w);
        DPTX_P2_ISSUE_A(a_ptr((kt + 1) & 1, 1));
      }
      asm volatile("s_barrier" ::: "memory");
      mma_tile<DT, TM, TN, RELU_A, PL, BK / 16, XT>(sa, sb, Q, 128 * 128, 0, wn, lr, lh, acc);   // slot 2
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // what group 1 reads in its next slot has landed
      asm volatile("s_barrier" ::: "memory");
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      const char* sa = a_ptr(kt & 1, 1);
      const char* sb = w_ptr(kt & 1);
      mma_tile<DT, TM, TN, RELU_A, PL, BK / 16, XT>(sa, sb, Q, 128 * 128, 0, wn, lr, lh, acc);   // slot 1
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // its DMA of the previous slot 2 (group 0's rows of THIS tile)
      asm volatile("s_barrier" ::: "memory");
      if (kt + 1 < nk) DPTX_P2_ISSUE_A(a_ptr((kt + 1) & 1, 0));   // slot 2
      asm volatile("s_barrier" ::: "memory");
    }
  }
#undef DPTX_P2_ISSUE_W
#undef DPTX_P2_ISSUE_A
#undef DPTX_P2_NEXT
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // all waves finished reading the stages: re-use LDS for the C tile
  epilogue<DT, BM, BN, TM, TN, 2, NT, 1>(p, smem, m0, n0, wm, wn, lr, lh, tid, acc);
#endif
}



template <int DT, int BM, int BN, int WAVES_M, int WAVES_N, bool A_FP32>
__global__ __launch_bounds__(256, 2) void gemm_reg_kernel(const GemmParams p) {
  static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
  constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
  constexpr int A_PASSES = BM / 32, B_PASSES = BN / 32;
  constexpr int A_REGS = A_FP32 ? 2 : 1;
  constexpr int STAGE_BYTES = (BM + BN) * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;
  const int lr = lane & 31, lh = lane >> 5;

  // block -> tile: XCD x = blockIdx % 8 (observed dispatch placement; only speed depends on it) owns the
  // (x / xcd_n)-th m-slice and (x % xcd_n)-th n-slice of the tile grid; inside a slice n runs fastest so
  // consecutive blocks of one XCD share their A rows.
  const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM;
  int m0, n0;
  {
    const int x = (int)blockIdx.x & 7, l = (int)blockIdx.x >> 3;
    const int tn_per = tiles_n / p.xcd_n, tm_per = (tiles_m + p.xcd_m - 1) / p.xcd_m;
    const int mt = (x / p.xcd_n) * tm_per + l / tn_per;
    const int nt = (x % p.xcd_n) * tn_per + l % tn_per;
    if (mt >= tiles_m || l >= tm_per * tn_per) return;
    m0 = mt * BM;
    n0 = nt * BN;
  }

  const int kc = tid & 7, r0 = tid >> 3;
  const bool big_m = p.M >= (1 << 23);
  int a_iy0[A_PASSES], a_ix0[A_PASSES];
  long long a_base[A_PASSES];
#pragma unroll
  for (int i = 0; i < A_PASSES; ++i) {
    const int m = m0 + r0 + 32 * i;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    int rem, ox;
    const int img = row_div(mm, p.a_rpi, p.a_rpi_rcp, big_m, rem);
    const int oy = row_div(rem, p.Wout, p.wout_rcp, big_m, ox);
    a_iy0[i] = ok ? oy * p.stride - p.pad_t : -0x40000000;
    a_ix0[i] = ox * p.stride - p.pad_l;
    a_base[i] = (long long)img * p.a_img_stride + p.a_off + ((long long)a_iy0[i] * p.Win + a_ix0[i]) * p.a_pix_stride + kc * 8;
  }
  const char* __restrict__ Ab = (const char*)p.A;
  const uint16_t* __restrict__ Wb = (const uint16_t*)p.W;
  long long w_off[B_PASSES];
#pragma unroll
  for (int j = 0; j < B_PASSES; ++j) w_off[j] = (long long)(n0 + r0 + 32 * j) * p.ldw + kc * 8;

  u32x4_t ra[A_PASSES * A_REGS];
  u32x4_t rb[B_PASSES];
  int ky = 0, kx = 0, c0 = 0;
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};

#define DPTX_LOAD_TILE(K0)                                                                           \
  do {                                                                                               \
    const long long tap_ = (long long)(ky * p.Win + kx) * p.a_pix_stride + c0;                       \
    _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                           \
      const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;                                              \
      const bool valid = ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);       \
      const long long e = a_base[i] + tap_;                                                          \
      if constexpr (A_FP32) {                                                                        \
        u32x4_t lo = zero4, hi = zero4;                                                              \
        if (valid) {                                                                                 \
          const u32x4_t* src = (const u32x4_t*)(Ab + e * 4);                                         \
          lo = src[0];                                                                               \
          hi = src[1];                                                                               \
        }                                                                                            \
        ra[2 * i] = lo;                                                                              \
        ra[2 * i + 1] = hi;                                                                          \
      } else {                                                                                       \
        u32x4_t v = zero4;                                                                           \
        if (valid) v = *(const u32x4_t*)(Ab + e * 2);                                                \
        ra[i] = v;                                                                                   \
      }                                                                                              \
    }                                                                                                \
    _Pragma("unroll") for (int j = 0; j < B_PASSES; ++j) rb[j] = *(const u32x4_t*)(Wb + w_off[j] + (K0)); \
    c0 += BK;                                                                                        \
    if (c0 >= p.Cin) {                                                                               \
      c0 = 0;                                                                                        \
      if (++kx == p.ksz) { kx = 0; ++ky; }                                                           \
    }                                                                                                \
  } while (0)

#define DPTX_STORE_TILE(BUF)                                                                         \
  do {                                                                                               \
    char* sa_ = smem + (BUF) * STAGE_BYTES;                                                          \
    char* sb_ = sa_ + BM * 128;                                                                      \
    _Pragma("unroll") for (int i = 0; i < A_PASSES; ++i) {                                           \
      const int row = r0 + 32 * i;                                                                   \
      u32x4_t v;                                                                                     \
      if constexpr (A_FP32) {                                                                        \
        const u32x4_t lo = ra[2 * i], hi = ra[2 * i + 1];                                            \
        v.x = T16<DT>::pack2(__uint_as_float(lo.x), __uint_as_float(lo.y));                          \
        v.y = T16<DT>::pack2(__uint_as_float(lo.z), __uint_as_float(lo.w));                          \
        v.z = T16<DT>::pack2(__uint_as_float(hi.x), __uint_as_float(hi.y));                          \
        v.w = T16<DT>::pack2(__uint_as_float(hi.z), __uint_as_float(hi.w));                          \
      } else {                                                                                       \
        v = ra[i];                                                                                   \
      }                                                                                              \
      if (p.a_relu) v = relu8(v);                                                                    \
      *(u32x4_t*)(sa_ + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4)) = v;                             \
    }                                                                                                \
    _Pragma("unroll") for (int j = 0; j < B_PASSES; ++j) {                                           \
      const int row = r0 + 32 * j;                                                                   \
      *(u32x4_t*)(sb_ + row * 128 + ((kc ^ ((row >> 1) & 7)) << 4)) = rb[j];                         \
    }                                                                                                \
  } while (0)

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  DPTX_LOAD_TILE(0);
  DPTX_STORE_TILE(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = (kt + 1) < nk;
    if (more) DPTX_LOAD_TILE((kt + 1) * BK);
    const char* sa = smem + (kt & 1) * STAGE_BYTES;
    mma_tile<DT, TM, TN, false, 1>(sa, sa + BM * 128, 0, 0, wm, wn, lr, lh, acc);
    if (more) DPTX_STORE_TILE((kt + 1) & 1);
    __syncthreads();
  }
#undef DPTX_LOAD_TILE
#undef DPTX_STORE_TILE
  epilogue<DT, BM, BN, TM, TN>(p, smem, m0, n0, wm, wn, lr, lh, tid, acc);
}

// --------------------------------------------------------------------------------- dispatch
template <int BM, int BN, int PL>
constexpr size_t gemm_smem_bytes() {
  constexpr size_t stage = 2 * (size_t)PL * (BM + BN) * 128;
  constexpr size_t ct_full = (size_t)BM * (BN + 4) * 4;
  constexpr size_t ct = (ct_full > 160 * 1024 ? (size_t)64 * (BN + 4) * 4 : ct_full)  // slab epilogue (2 wave rows x 32)
                        + (size_t)BM * 8;  // + the (mu, rstd) row table of the LayerNorm fold
  return stage > ct ? stage : ct;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per (kernel, device) --
// one process may hold engines on several GPUs (kernels.h ensure_dyn_smem, defined in misc.hip)
template <typename K>
static void set_smem_attr(K k, size_t smem) { ensure_dyn_smem((const void*)k, smem); }

// A/B experiments: DPTX_GEMM=reg forces the register-staged kernel; DPTX_XCD=0 disables the 2-D XCD
// partition of the tile grid.
static int gemm_variant() {
  static int v = -1;
  if (v < 0) {
    const char* s = getenv("DPTX_GEMM");
    v = (s && s[0] == 'r') ? 1 : 0;
  }
  return v;
}
static int xcd_partition_enabled() {
  static int v = -1;
  if (v < 0) {
    const char* s = getenv("DPTX_XCD");
    v = (s && s[0] == '0') ? 0 : 1;
  }
  return v;
}

// Chooses the XCD grid (xm x xn, xm*xn = 8) for a tile grid: XCD (xi, xj) owns the m-tiles of slice xi
// and the n-tiles of slice xj, so per XCD only A/xm and W/xn are touched.  Objective: HBM/MALL-side
// traffic xn*A_bytes + xm*W_bytes, subject to the XCD's W slice fitting comfortably in its 4 MB L2
// (otherwise W is re-fetched for every m-tile) and xn dividing the n-tile count.
static void choose_xcd_grid(const GemmParams& p, int tiles_m, int tiles_n, int& xm, int& xn) {
  xm = 8; xn = 1;
  if (!xcd_partition_enabled()) return;
  const double a_bytes = (double)p.M * p.K * 2.0, w_bytes = (double)p.N * p.K * 2.0;
  double best = -1.0;
  const int cand[4] = {1, 2, 4, 8};
  for (int c = 0; c < 4; ++c) {
    const int n = cand[c], m = 8 / n;
    if (tiles_n % n != 0 || tiles_m < m) continue;
    const double w_slice = w_bytes / n;
    double cost = n * a_bytes + m * w_bytes;
    if (w_slice > 2.0e6) cost += (double)tiles_m / m * w_slice * 8.0;  // L2 thrash: W slice re-read per m-tile
    if (best < 0.0 || cost < best) { best = cost; xm = m; xn = n; }
  }
}


template <int DT, int PL, int BM, int BN, int WM_, int WN_, int PLE = PL, int XT = 3>
static hipError_t launch_cfg(const GemmParams& p, hipStream_t stream) {
  const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
  GemmParams q = p;
  choose_xcd_grid(p, tiles_m, tiles_n, q.xcd_m, q.xcd_n);
  const int tiles = 8 * ((tiles_m + q.xcd_m - 1) / q.xcd_m) * (tiles_n / q.xcd_n);
  constexpr size_t smem = gemm_smem_bytes<BM, BN, PL>();
  const bool glds_ok = !p.a_fp32 && p.a_bytes > 0 && p.a_bytes < (1ll << 31) && (long long)p.N * p.ldw * 2 < (1ll << 31) &&
                       p.M < (1 << 23);
  if (PL == 2 && !glds_ok) return hipErrorInvalidValue;  // the 3-pass mode exists only on the direct-to-LDS path
  if constexpr (DT != DT_FP8 && PL == 1 && BM == 256 && BN == 256) {
    constexpr bool pp = true;
    if (pp && glds_ok) {  // ping-pong schedule of the two wave groups (gemm_pp_kernel)
      // persistent launch: at most 32 blocks per XCD (one per CU), each looping over its XCD slice's tiles; the epilogue's
      // LDS tile sits behind stage 0 (which receives the next tile's first k-tile meanwhile).  DPTX_PERSIST=0: one block per
      // tile (A/B runs; same kernel, same results); DPTX_PERSIST=n: n blocks per XCD
      static int per_xcd = -1;
      if (per_xcd < 0) { const char* e = getenv("DPTX_PERSIST"); per_xcd = e ? atoi(e) : 32; }
      const int ltot = tiles / 8;
      const int grid = per_xcd > 0 && ltot > per_xcd && !(p.debug_flags & 2) ? 8 * per_xcd : tiles;
      // (wave-private staging: 8 waves x 32 rows x 72 floats; block-wide staging of the row_stats launches: 64 x 260 floats)
      constexpr size_t smem_pp = (size_t)256 * 256 + (size_t)8 * 32 * 72 * 4 + (size_t)BM * 8 + (size_t)BM * 64;  // + the statistics records
      static_assert((size_t)8 * 32 * 72 * 4 >= (size_t)64 * (BN + 4) * 4, "the block-wide slab fits too");
      static_assert(smem_pp >= smem && smem_pp <= 160 * 1024, "stage 0 + the epilogue's slab and row table");
      auto go = [&](auto k) {
        set_smem_attr(k, smem_pp);
        hipLaunchKernelGGL(k, dim3(grid), dim3(512), smem_pp, stream, q);
      };
      // plain epilogue (bias, LayerNorm fold, activation, one 16-bit plane): the transposed-accumulator form that stores
      // straight from the registers.  DPTX_DIRECT=0: the staged epilogue everywhere (A/B runs; results agree bit for bit)
      static int direct_on = -1;
      if (direct_on < 0) { const char* e = getenv("DPTX_DIRECT"); direct_on = e ? atoi(e) : 2; }
      const bool direct_any = direct_on && !(p.debug_flags & 1) && PLE == 1 && !p.bias_per_img && p.c_rpi == 0x7fffffff &&
                              p.C8 == nullptr && p.gn_part == nullptr && p.out_scale == 0.f;
      const bool direct = direct_any && p.R1 == nullptr && p.R2 == nullptr && p.row_stats == nullptr && !p.c_fp32 && p.C16 == nullptr;
      // round 6: residuals and the producer side of the LayerNorm fold in the direct form (epilogue_direct RES 1, 2): proj / fc2
      // on the 16-bit token stream, the second convolution of a RCU with and without the fusion path.
      // DPTX_DIRECT=1: round 5's set only
      const bool direct_res = direct_any && direct_on >= 2 && p.R1 != nullptr && !p.a_relu && p.ln_stats == nullptr;
      const int res = !direct_res ? 0
                      : (!p.r1_fp32 && !p.c_fp32 && p.C16 == nullptr && p.R2 == nullptr) ? 1
                      : (!p.r1_fp32 && !p.c_fp32 && p.C16 == nullptr && p.R2 != nullptr && !p.r2_fp32 && !p.r2_bcast && p.row_stats == nullptr) ? 2
                      : 0;
      if constexpr (PLE == 1) {
        if (direct) {
          if (p.a_relu) go(gemm_pp_kernel<DT, true, 1, 1>);
          else go(gemm_pp_kernel<DT, false, 1, 1>);
          return hipGetLastError();
        }
        if (res == 1) { go(gemm_pp_kernel<DT, false, 1, 2>); return hipGetLastError(); }
        if (res == 2) { go(gemm_pp_kernel<DT, false, 1, 3>); return hipGetLastError(); }
      }
      if (p.a_relu) go(gemm_pp_kernel<DT, true, PLE>);
      else go(gemm_pp_kernel<DT, false, PLE>);
      return hipGetLastError();
    }
  }
  if constexpr (DT != DT_FP8 && PL == 2 && PLE == 2 && BM == 128 && BN == 128 && WM_ == 2 && WN_ == 4) {
    // the two-plane 128x128 tile: ping-pong schedule of the two wave groups (gemm_pp2_kernel).  DPTX_PP2=0: lockstep loop
    static int pp2 = -1;
    if (pp2 < 0) { const char* e = getenv("DPTX_PP2"); pp2 = e ? atoi(e) : 1; }
    // (not with the pre-activation on the A fragments: its VALU work sits inside the MFMA slot, and the two RCU convolutions
    // that have it were 8-9 % SLOWER than with the lockstep loop: profiles/r04_experiments.md)
    if (pp2 && !(p.debug_flags & 4) && p.K >= 2 * BK && !p.a_relu) {
      auto k = gemm_pp2_kernel<DT, false, XT>;
      set_smem_attr(k, smem);
      hipLaunchKernelGGL(k, dim3(tiles), dim3(512), smem, stream, q);
      return hipGetLastError();
    }
  }
  if constexpr (DT == DT_FP8) {  // fp8 operands: direct-to-LDS path only, pre-activation is the producer's job
    if (!glds_ok || p.a_relu) return hipErrorInvalidValue;
    auto k = gemm_glds_kernel<DT, BM, BN, WM_, WN_, false, PL, PLE>;
    set_smem_attr(k, smem);
    hipLaunchKernelGGL(k, dim3(tiles), dim3(64 * WM_ * WN_), smem, stream, q);
    return hipGetLastError();
  } else
  if (glds_ok && (PL == 2 || gemm_variant() != 1)) {
    if (p.a_relu) {
      auto k = gemm_glds_kernel<DT, BM, BN, WM_, WN_, true, PL, PLE, XT>;
      set_smem_attr(k, smem);
      hipLaunchKernelGGL(k, dim3(tiles), dim3(64 * WM_ * WN_), smem, stream, q);
    } else {
      auto k = gemm_glds_kernel<DT, BM, BN, WM_, WN_, false, PL, PLE, XT>;
      set_smem_attr(k, smem);
      hipLaunchKernelGGL(k, dim3(tiles), dim3(64 * WM_ * WN_), smem, stream, q);
    }
  } else if constexpr (WM_ * WN_ != 4 || PLE != PL) {
    return hipErrorInvalidValue;  // the 8-wave tile and the two-plane epilogue exist only on the direct-to-LDS path
  } else if constexpr (PL == 1) {
    constexpr size_t smem1 = gemm_smem_bytes<BM, BN, 1>();
    if (p.a_fp32) {
      // fp32 A exists for dense GEMMs with wide outputs only (the 256-row tiles serve N = 32 / 64 convolutions; their
      // fp32-staging variant would hold 64 staging + 64 accumulator + 48 fragment registers and spill)
      if constexpr (BM == 256) return hipErrorInvalidValue;
      else {
        auto k = gemm_reg_kernel<DT, BM, BN, WM_, WN_, true>;
        set_smem_attr(k, smem1);
        hipLaunchKernelGGL(k, dim3(tiles), dim3(256), smem1, stream, q);
      }
    } else {
      auto k = gemm_reg_kernel<DT, BM, BN, WM_, WN_, false>;
      set_smem_attr(k, smem1);
      hipLaunchKernelGGL(k, dim3(tiles), dim3(256), smem1, stream, q);
    }
  }
  return hipGetLastError();
}

template <int DT, int PL, int PLE = PL, int XT = 3>
static hipError_t launch_dt(const GemmParams& p, hipStream_t stream) {
  // tile choice: widest tile that still yields >= ~2 blocks per CU (256 CUs); N must divide.
  const long long m128 = (p.M + 127) / 128, m256 = (p.M + 255) / 256;
  // cu_share: the part of the chip this launch can count on -- 1/n when the forward runs as n sub-batches on n streams (the
  // other streams' launches occupy the rest), so that a half-batch GEMM is tiled for half the CUs instead of being judged
  // too small for the wide tiles: CU slots and tile-count thresholds scale with it
  const double sh = p.cu_share > 0.f ? (double)p.cu_share : 1.0;
  const double shs = p.cu_share_small > 0.f ? (double)p.cu_share_small : sh;  // for the narrow-tile thresholds
  const long long cu1 = (long long)(256 * sh + 0.5), cu2 = (long long)(512 * sh + 0.5);  // slots at 1 / 2 blocks per CU
  static int forced = -1;  // DPTX_TILE=128 disables the 256x256 tile; 12864 / 6464 force a tile shape (tools/gemm_bench.py)
  if (forced < 0) { const char* t = getenv("DPTX_TILE"); forced = t ? atoi(t) : 0; }
  {
    if (forced == 128128 && p.N % 128 == 0) return launch_cfg<DT, PL, 128, 128, 2, 2, PLE, XT>(p, stream);
    if (forced == 12864 && p.N % 64 == 0) return launch_cfg<DT, PL, 128, 64, 2, 2, PLE, XT>(p, stream);
    if (forced == 6464 && p.N % 64 == 0) return launch_cfg<DT, PL, 64, 64, 2, 2, PLE, XT>(p, stream);
  }
  // 256x256 (8 waves, 1 block/CU; 16-bit modes: the ping-pong kernel): half the DMA issues and 3/4 of the LDS reads per MFMA
  // of the 128x128 tile, but no second block to hide prologue/epilogue and a coarser tail.  Chosen when its estimated
  // efficiency wins: fill of the last round of CUs (256 slots) x 1.5 (the per-tile advantage at K >= 512: 1.25 with round 2's
  // kernel; the persistent tile loop and the register-direct epilogue of round 3 moved the optimum, profiles/r03_experiments.md)
  // against the fill of the 128x128 grid (512 slots).  That picks
  // it for the ViT GEMMs, patch-embed and the 3x3 convs at 1/4 resolution and keeps 128x128 for the small maps.
  if constexpr (PL == 1) {
    const bool glds_ok = !p.a_fp32 && p.a_bytes > 0 && p.a_bytes < (1ll << 31) && p.M < (1 << 23) && gemm_variant() != 1;
    if (forced != 128 && glds_ok && p.N % 256 == 0 && p.K >= 512) {
      const long long t256 = m256 * (p.N / 256), r256 = (t256 + cu1 - 1) / cu1;
      const long long t128 = m128 * (p.N / 128), r128 = (t128 + cu2 - 1) / cu2;
      const double fill256 = (double)t256 / (double)(r256 * cu1), fill128 = (double)t128 / (double)(r128 * cu2);
      static double adv = -1.0;  // DPTX_PP_ADV: the per-tile advantage assumed for the 256x256 kernel (A/B runs)
      if (adv < 0.0) { const char* e = getenv("DPTX_PP_ADV"); adv = e ? atof(e) : 1.5; }
      if (t256 >= 200 * sh && fill256 * adv >= fill128) return launch_cfg<DT, PL, 256, 256, 2, 4, PLE, XT>(p, stream);
    }
  }
  if constexpr (PL == 2) {
    // 3-MFMA modes are one block per CU (two planes of two stages = 128 KB of LDS); eight waves (2 x 4, wave tile
    // 64 x 32) instead of four put two waves on every SIMD, so that one's fragment reads overlap the other's MFMAs:
    // GEMM family 32.7 -> 30.5 ms per fp16x3 forward (profiles/r02_experiments.md)
    if (p.N % 128 == 0 && m128 * (p.N / 128) >= 200 * shs) return launch_cfg<DT, PL, 128, 128, 2, 4, PLE, XT>(p, stream);
  }
  // (row_stats -- the producer side of the LayerNorm fold -- reduces 128-column blocks inside a tile: never narrower tiles)
  // 128x128 from 256 tiles up (one block on every CU): at 288 tiles (M = 18432, N = 256) it still beats 576 tiles of
  // 128x64 by 2..10 %, whose second round is nearly empty
  if (p.N % 128 == 0 && (m128 * (p.N / 128) >= 256 * shs || p.row_stats != nullptr)) return launch_cfg<DT, PL, 128, 128, 2, 2, PLE, XT>(p, stream);
  if (p.N == 32) return launch_cfg<DT, PL, 256, 32, 4, 1, PLE, XT>(p, stream);
  if (PL == 1 && p.N % 64 == 0 && p.N < 128 && m256 * (p.N / 64) >= 448 * shs) return launch_cfg<DT, PL, 256, 64, 4, 1, PLE, XT>(p, stream);
  if (p.N % 64 == 0 && m128 * (p.N / 64) >= 448 * shs) return launch_cfg<DT, PL, 128, 64, 2, 2, PLE, XT>(p, stream);
  if (p.N % 64 == 0) return launch_cfg<DT, PL, 64, 64, 2, 2, PLE, XT>(p, stream);
  return hipErrorInvalidValue;
}


// per-translation-unit entry points (launch_gemm dispatches to them)
hipError_t launch_gemm_16(int dt, const GemmParams& p, hipStream_t stream);   // gemm.hip:     DT_BF16 one plane (fp16: next line)
hipError_t launch_gemm_fp16(const GemmParams& p, hipStream_t stream);         // gemm_fp16.hip: DT_FP16, one plane
hipError_t launch_gemm_fp16e(const GemmParams& p, hipStream_t stream);        // gemm_fp16e.hip: DT_FP16, one plane, two-plane epilogue
hipError_t launch_gemm_x3(int dt, const GemmParams& p, hipStream_t stream);   // gemm_x3.hip:  DT_BF16 / DT_FP16, hi/lo planes
hipError_t launch_gemm_x2(const GemmParams& p, hipStream_t stream);           // gemm_x2.hip:  DT_FP16, hi/lo planes, A hi only (2 MFMAs)
hipError_t launch_gemm_fp8(const GemmParams& p, hipStream_t stream);          // gemm_fp8.hip: e4m3

}  // namespace dptx
