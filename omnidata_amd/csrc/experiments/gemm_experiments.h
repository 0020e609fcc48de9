#pragma once
// experiments/gemm_experiments.h -- schedule experiments of round 2 (profiles/r02_experiments.md): the 256x128 three-stage
// kernel, the halo-resident 3x3 convolution and the phased ("8-phase") 256x256 kernel.  All correct, all measured slower
// than gemm_pp_kernel; NOT part of the default build (build.py compiles them only with DPTX_CXXFLAGS=-DDPTX_EXPERIMENTS,
// which also switches on the DPTX_PP / DPTX_HALO environment variables that select them).  Included by gemm_impl.h.

// ------------------------------------------------------------------- 256x128 three-stage kernel (experiment, DPTX_PP=7)
// The probes of the ping-pong kernel say its DMA side is latency x capacity: ~1800 cycles from issue to landed, one 64-KB
// k-tile in flight per CU.  Here the tile is 256 x 128 (48 KB per k-tile) and the LDS holds THREE stages (144 KB): two
// k-tiles = 96 KB are in flight while the third is multiplied.  One barrier per k-tile (schedule of gemm_pp_kernel VAR 3):
// tile kt is complete when iteration kt starts; group 0 (rows 0..127) issues its half of tile kt+2 and goes into its MFMAs,
// group 1 (rows 128..255; s_setprio: it is the younger half of the workgroup) multiplies first and issues its half
// afterwards; every wave waits with vmcnt(6) -- its six pieces of tile kt+2 may fly, tile kt+1's have landed.
// Waves 4 (M) x 2 (N), wave tile 64 x 64.
struct P3Frags { u32x4_t a[2], b[2]; };
template <int DT, bool RELU_A>
__global__ __launch_bounds__(512, 2) void gemm_p3_kernel(const GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 256, BN = 128, NT = 512, TM = 2, TN = 2, PL = 1;
  constexpr int A_BYTES = 256 * 128, STAGE = (256 + 128) * 128;  // 48 KB
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;   // 4 x 2 waves
  const int g = wave >> 2, q = wave & 3;     // group (128-row half) and wave inside it
  const int lr = lane & 31, lh = lane >> 5;

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
  // loader: wave q of group g owns rows g*128 + 32 i + 8 q + (lane >> 3) of A (i = 0..3) and g*64 + 32 j + 8 q + (lane >> 3)
  // of W (j = 0, 1); chunk kc = lane & 7
  const int kc = lane & 7, r0 = 8 * q + (lane >> 3);
  const int sc = kc ^ ((r0 >> 1) & 7);
  int a_iy0[4], a_ix0[4];
  unsigned a_off[4], w_off[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + g * 128 + r0 + 32 * i;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    int rem, ox;
    const int img = row_div(mm, p.a_rpi, p.a_rpi_rcp, false, rem);
    const int oy = row_div(rem, p.Wout, p.wout_rcp, false, ox);
    a_iy0[i] = ok ? oy * p.stride - p.pad_t : -0x40000000;
    a_ix0[i] = ox * p.stride - p.pad_l;
    const long long e = (long long)img * p.a_img_stride + p.a_off + ((long long)a_iy0[i] * p.Win + a_ix0[i]) * p.a_pix_stride + sc * 8;
    a_off[i] = (unsigned)(ok ? e * 2 : 0);
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) w_off[j] = (unsigned)(((long long)(n0 + g * 64 + r0 + 32 * j) * p.ldw + sc * 8) * 2);
  const int w_bytes = (int)((long long)p.N * p.ldw * 2);
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, w_bytes, 0x00020000);

  int ky = 0, kx = 0, c0 = 0;  // tap / channel offset of the k-tile this wave loads next
  // this wave's six pieces of the k-tile at (ky, kx, c0) into stage ST; advances the tap
#define DPTX_P3_ISSUE(ST)                                                                                          \
  do {                                                                                                             \
    char* st_ = smem + (ST) * STAGE + q * 1024;                                                                    \
    const unsigned tap_ = (unsigned)(((ky * p.Win + kx) * p.a_pix_stride + c0) * 2);                               \
    const unsigned wk_ = (unsigned)((((ky * p.ksz + kx) * p.Cin) + c0) * 2);                                       \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                  \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void*)(st_ + A_BYTES + (g * 64 + 32 * j) * 128), \
                                               16, w_off[j] + wk_, 0, 0, 0);                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                                \
      const int iy = a_iy0[i] + ky, ix = a_ix0[i] + kx;                                                            \
      const bool valid = ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);                     \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (__attribute__((address_space(3))) void*)(st_ + (g * 128 + 32 * i) * 128), 16, \
                                               valid ? a_off[i] + tap_ : OOB, 0, 0, 0);                            \
    }                                                                                                              \
    if (p.k_tap_fast) {                                                                                            \
      if (++kx == p.ksz) { kx = 0; if (++ky == p.ksz) { ky = 0; c0 += BK; } }                                      \
    } else {                                                                                                       \
      c0 += BK;                                                                                                    \
      if (c0 >= p.Cin) { c0 = 0; if (++kx == p.ksz) { kx = 0; ++ky; } }                                            \
    }                                                                                                              \
  } while (0)

  auto read = [&](P3Frags& f, const char* st, int ks) {
    const int chunk = 2 * ks + lh;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = wm * 64 + i * 32 + lr;
      f.a[i] = *(const u32x4_t*)(st + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wn * 64 + j * 32 + lr;
      f.b[j] = *(const u32x4_t*)(st + A_BYTES + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
    }
  };
  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto mma = [&](P3Frags& f) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (RELU_A) f.a[i] = relu8(f.a[i]);
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = T16<DT>::mfma32(f.a[i], f.b[j], acc[i][j]);
    }
  };
  P3Frags f0, f1;
  auto mma_tile3 = [&](const char* st) {
    read(f0, st, 0);
    read(f1, st, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(f0);
    __builtin_amdgcn_sched_barrier(0);
    read(f0, st, 2);
    __builtin_amdgcn_sched_barrier(0);
    mma(f1);
    __builtin_amdgcn_sched_barrier(0);
    read(f1, st, 3);
    __builtin_amdgcn_sched_barrier(0);
    mma(f0);
    __builtin_amdgcn_sched_barrier(0);
    mma(f1);
  };

  const int nk = p.K / BK;
  // prologue: tiles 0 and 1
  DPTX_P3_ISSUE(0);
  if (nk > 1) DPTX_P3_ISSUE(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int s_rd = 0, s_wr = 2;
  if (g == 0) {
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 2 < nk;
      if (more) DPTX_P3_ISSUE(s_wr);
      mma_tile3(smem + s_rd * STAGE);
      s_rd = s_rd == 2 ? 0 : s_rd + 1;
      s_wr = s_wr == 2 ? 0 : s_wr + 1;
      if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
    }
  } else {
    __builtin_amdgcn_s_setprio(2);
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 2 < nk;
      mma_tile3(smem + s_rd * STAGE);
      if (more) DPTX_P3_ISSUE(s_wr);
      s_rd = s_rd == 2 ? 0 : s_rd + 1;
      s_wr = s_wr == 2 ? 0 : s_wr + 1;
      if (more) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
    }
    __builtin_amdgcn_s_setprio(0);
  }
#undef DPTX_P3_ISSUE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  epilogue<DT, BM, BN, TM, TN, PL, NT, 1>(p, smem, m0, n0, wm, wn, lr, lh, tid, acc);
#endif
}

// ------------------------------------------------------------------- halo-resident 3x3 convolution
// 3x3 / stride 1 / pad 1 convolutions on maps whose width is a multiple of 32 and height a multiple of 8 (the 1/4- and
// 1/2-resolution maps of the decoder).  The implicit GEMM above fetches every input pixel nine times, once per tap, as part
// of nine different A tiles; here a block owns 8 rows x 32 pixels of ONE image and 256 output channels, keeps the 10 x 34
// input halo of a 64-channel chunk LDS-resident (43.5 KB) and runs the nine taps out of it: per 64-channel chunk the block
// moves 43.5 KB of A and 9 x 32 KB of W through the LDS-DMA path instead of 9 x 64 KB (1.7x fewer bytes per flop).
//
// k order: chunk-major, taps inside a chunk = GemmParams::k_tap_fast, which launch_gemm forces for every shape this
// kernel accepts, so that the result is bit-identical to the implicit-GEMM kernels' (small batches fall back to them).
//
// Same wave layout and ping-pong schedule as gemm_pp_kernel (group = 4 output rows x 32 pixels x 256 channels); group 0
// issues the eight W pieces of the next (tap, chunk), group 1 the halo of the NEXT chunk, two of its 43 pieces per wave
// and tap, which therefore have most of a chunk to land.
template <int DT, bool RELU_A>
__global__ __launch_bounds__(512, 2) void gemm_halo_kernel(const GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 256, BN = 256, NT = 512, TM = 4, TN = 2, PL = 1;
  constexpr int SLABS = TM;
  constexpr int HC = 34, HPIX = 10 * HC, HPIECES = (HPIX + 7) / 8;  // 340 halo pixels in 43 pieces of 8 rows
  constexpr int W_BYTES = 256 * 128, HALO_BYTES = 44 * 1024;        // 344 rows x 128 B = 44032 <= 45056
  constexpr int HSLOTS = (HPIECES + 3) / 4;                         // pieces per wave of group 1: 11
  extern __shared__ __attribute__((aligned(16))) char smem[];       // W x 2 | halo x 2 = 152 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3, wq = wave & 3;
  const int lr = lane & 31, lh = lane >> 5;
  auto w_ptr = [&](int b) -> char* { return smem + b * W_BYTES; };
  auto h_ptr = [&](int b) -> char* { return smem + 2 * W_BYTES + b * HALO_BYTES; };

  const int tx_n = p.Win >> 5, tpi = tx_n * (p.Hin >> 3);
  const int tiles_n = p.N / BN, tiles_m = (p.M / p.a_rpi) * tpi;
  int img, y0, x0, n0;
  {
    const int x = (int)blockIdx.x & 7, l = (int)blockIdx.x >> 3;
    const int tn_per = tiles_n / p.xcd_n, tm_per = (tiles_m + p.xcd_m - 1) / p.xcd_m;
    const int mt = (x / p.xcd_n) * tm_per + l / tn_per;
    const int nt = (x % p.xcd_n) * tn_per + l % tn_per;
    if (mt >= tiles_m || l >= tm_per * tn_per) return;
    img = mt / tpi;
    const int r = mt - img * tpi, ty = r / tx_n;
    y0 = ty * 8;
    x0 = (r - ty * tx_n) * 32;
    n0 = nt * BN;
  }
  const int m_base = (img * p.Hin + y0) * p.Win + x0;

  // loader offsets, one array for both roles (the groups never meet in this code):
  //   group 1, wave wq: halo pieces id = 4 s + wq (s = 0..10); a piece is 8 halo pixels x 128 B, lane (lane>>3, lane&7)
  //     owns chunk kc of halo pixel hp = 8 id + (lane >> 3); pixels outside the image (and the 4 rows past 339) read zeros
  //   group 0: W rows r0 + 32 j (j = 0..7) of the 256-row tile, chunk kc (as in gemm_pp_kernel)
  unsigned offs[HSLOTS];
  if (wm == 1) {
    const int kc = lane & 7;
#pragma unroll
    for (int s_ = 0; s_ < HSLOTS; ++s_) {
      const int id = 4 * s_ + wq, hp = 8 * id + (lane >> 3);
      const int hy = hp / HC, hx = hp - hy * HC;
      const int y = y0 - 1 + hy, xx = x0 - 1 + hx;
      const bool ok = id < HPIECES && hp < HPIX && (unsigned)y < (unsigned)p.Hin && (unsigned)xx < (unsigned)p.Win;
      const int sc = kc ^ ((hp >> 1) & 7);
      const long long e = (long long)img * p.a_img_stride + p.a_off + ((long long)y * p.Win + xx) * p.a_pix_stride + sc * 8;
      offs[s_] = ok ? (unsigned)(e * 2) : OOB;
    }
  } else {
    const int t = tid & 255, kc = t & 7, r0 = t >> 3;
    const int sc = kc ^ ((r0 >> 1) & 7);
#pragma unroll
    for (int j = 0; j < HSLOTS; ++j) offs[j] = j < 8 ? (unsigned)(((long long)(n0 + r0 + 32 * j) * p.ldw + sc * 8) * 2) : 0u;
  }
  const int w_bytes = (int)((long long)p.N * p.ldw * 2);
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, w_bytes, 0x00020000);

  // halo pieces of slots S0, S0 + 1 (static) of the chunk at channel C0 into halo buffer DST
#define DPTX_HALO_ISSUE_A(DST, S0, C0)                                                                             \
  do {                                                                                                             \
    _Pragma("unroll") for (int s_ = (S0); s_ < (S0) + 2; ++s_) {                                                   \
      if (s_ < HSLOTS && 4 * s_ + wq < HPIECES)                                                                    \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                                  \
            rsrcA, (__attribute__((address_space(3))) void*)((DST) + (4 * s_ + wq) * 1024), 16,                    \
            offs[s_ < HSLOTS ? s_ : 0] == OOB ? OOB : offs[s_ < HSLOTS ? s_ : 0] + (unsigned)((C0) * 2), 0, 0, 0); \
    }                                                                                                              \
  } while (0)
  // the eight W pieces of (tap TAP, channel chunk C0) into W buffer DST
#define DPTX_HALO_ISSUE_W(DST, TAP, C0)                                                                            \
  do {                                                                                                             \
    char* d_ = (DST) + wq * 1024;                                                                                  \
    const unsigned wk_ = (unsigned)(((TAP) * p.Cin + (C0)) * 2);                                                   \
    _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                                  \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void*)(d_ + 32 * j * 128), 16, \
                                               offs[j] + wk_, 0, 0, 0);                                            \
  } while (0)

  // fragment reads: A block i = output row 4 wm + i, lane lr = pixel; tap (ky, kx) shifts the halo pixel by ky*34 + kx
  // (the halo offsets of all nine unrolled taps are loop-invariant; left to itself hipcc hoists the 144 of them out of the
  // chunk loop and spills -- hp0 is made opaque once per tap so that they are recomputed, ~60 VALU per 32 MFMAs)
  int hp0 = wm * 4 * HC + lr;
  auto read = [&](PpFrags& f, const char* hb, const char* sb, int delta, int ks) {
    const int chunk = 2 * ks + lh;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int hp = hp0 + i * HC + delta;
      f.a[i] = *(const u32x4_t*)(hb + hp * 128 + ((chunk ^ ((hp >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = wn * 64 + j * 32 + lr;
      f.b[j] = *(const u32x4_t*)(sb + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
    }
  };
  PpFrags f0, f1;
  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  auto mma_tap = [&](const char* hb, const char* sb, int delta) {
    asm volatile("" : "+v"(hp0));
    read(f0, hb, sb, delta, 0);
    read(f1, hb, sb, delta, 1);
    __builtin_amdgcn_sched_barrier(0);
    pp_mma<DT, RELU_A>(f0, acc);
    __builtin_amdgcn_sched_barrier(0);
    read(f0, hb, sb, delta, 2);
    __builtin_amdgcn_sched_barrier(0);
    pp_mma<DT, RELU_A>(f1, acc);
    __builtin_amdgcn_sched_barrier(0);
    read(f1, hb, sb, delta, 3);
    __builtin_amdgcn_sched_barrier(0);
    pp_mma<DT, RELU_A>(f0, acc);
    __builtin_amdgcn_sched_barrier(0);
    pp_mma<DT, RELU_A>(f1, acc);
  };

  const int nch = p.Cin / BK;
  // prologue: the whole halo of chunk 0 (group 1) and W of (tap 0, chunk 0) (group 0)
  if (wm == 0) {
    DPTX_HALO_ISSUE_W(w_ptr(0), 0, 0);
  } else {
#pragma unroll
    for (int s0 = 0; s0 < HSLOTS + 1; s0 += 2) DPTX_HALO_ISSUE_A(h_ptr(0), s0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifndef DPTX_HALO_TWO_BARRIERS
  // ONE barrier per tap and overlapping MFMA slots (the schedule of gemm_pp_kernel VAR 3): group 0 issues the W pieces of
  // the next tap and goes straight into its MFMAs; group 1 multiplies first -- with the higher issue priority, or the older
  // half of the workgroup starves it -- and then issues its share of the next chunk's halo.
  if (wm == 0) {
    for (int cc = 0; cc < nch; ++cc) {
      const char* hb = h_ptr(cc & 1);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int wb = (cc + tap) & 1;
        if (tap < 8) DPTX_HALO_ISSUE_W(w_ptr(wb ^ 1), tap + 1, cc * BK);
        else if (cc + 1 < nch) DPTX_HALO_ISSUE_W(w_ptr(wb ^ 1), 0, (cc + 1) * BK);
        mma_tap(hb, w_ptr(wb), (tap / 3) * HC + tap % 3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
      }
    }
  } else {
    __builtin_amdgcn_s_setprio(2);
    for (int cc = 0; cc < nch; ++cc) {
      const char* hb = h_ptr(cc & 1);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int wb = (cc + tap) & 1;
        mma_tap(hb, w_ptr(wb), (tap / 3) * HC + tap % 3);
        if (2 * tap < HSLOTS && cc + 1 < nch) DPTX_HALO_ISSUE_A(h_ptr((cc + 1) & 1), 2 * tap, (cc + 1) * BK);
        if (tap == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next chunk's halo (issued in taps 0..5)
        asm volatile("s_barrier" ::: "memory");
      }
    }
  }
#else
  if (wm == 0) {
    for (int cc = 0; cc < nch; ++cc) {
      const char* hb = h_ptr(cc & 1);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int wb = (cc + tap) & 1;
        // slot 1: W of the next (tap, chunk)
        if (tap < 8) DPTX_HALO_ISSUE_W(w_ptr(wb ^ 1), tap + 1, cc * BK);
        else if (cc + 1 < nch) DPTX_HALO_ISSUE_W(w_ptr(wb ^ 1), 0, (cc + 1) * BK);
        asm volatile("s_barrier" ::: "memory");
        mma_tap(hb, w_ptr(wb), (tap / 3) * HC + tap % 3);        // slot 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
      }
    }
  } else {
    for (int cc = 0; cc < nch; ++cc) {
      const char* hb = h_ptr(cc & 1);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int wb = (cc + tap) & 1;
        mma_tap(hb, w_ptr(wb), (tap / 3) * HC + tap % 3);        // slot 1
        if (tap == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next chunk's halo (issued in taps 0..5)
        asm volatile("s_barrier" ::: "memory");
        if (2 * tap < HSLOTS && cc + 1 < nch) DPTX_HALO_ISSUE_A(h_ptr((cc + 1) & 1), 2 * tap, (cc + 1) * BK);  // slot 2
        asm volatile("s_barrier" ::: "memory");
      }
    }
  }
#endif
#undef DPTX_HALO_ISSUE_W
#undef DPTX_HALO_ISSUE_A
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  epilogue<DT, BM, BN, TM, TN, PL, NT, SLABS>(p, smem, m_base, n0, wm, wn, lr, lh, tid, acc, p.Win);
#endif
}

// ------------------------------------------------------------------- phased 256x256 kernel
// Four phases per k-tile, two barriers per phase, the two wave groups (wm = 0 / 1) one barrier apart:
//
//     group 0:  | reads + 2 DMA pieces | 8 MFMAs | reads + 2 DMA | 8 MFMAs | ...
//     group 1:            | 8 MFMAs | reads + 2 DMA pieces | 8 MFMAs | reads + ...
//
// so each SIMD always has one wave in its MFMA section while the other fetches its next fragments and issues its share of
// the LDS-DMA (2 of the 64 one-KB pieces of a k-tile per wave and phase, instead of bursts of 12 / 4 per k-tile in
// gemm_pp_kernel, whose fragment reads also sat in front of the MFMAs of the SAME wave).
//
// A k-tile is four 16-KB half-tiles: H0 = A rows 0..127, H1 = A rows 128..255, H2 = W rows 0..127, H3 = W rows 128..255.
// Wave (wm, wn) owns rows [64 wm, +64) of H0 AND of H1, columns [32 wn, +32) of H2 AND of H3 (epilogue<ILV>), and its 4 x 2
// accumulator blocks are visited as (top, j0), (top, j1), (bottom, j1), (bottom, j0): phase 0 reads H0 + H2, phase 1 H3,
// phase 2 H1, phase 3 H2 again.  Tile t+1 is staged in the same order -- H0, H2, H3, H1 in phases 0..3 of tile t --
// three (H0: four) phases before it is read, so the DMA of the two most recent steps stays in flight across the
// barriers: every wave waits at the end of its MFMA section, group 0 with vmcnt(4), group 1 (whose wait comes one barrier
// later) with vmcnt(2), and vmcnt(0) only in the last k-tile.  A half-tile buffer is re-staged at least two phases after
// its last fragment read (H2: read in phase 3 of tile t-1, written in phase 1 of tile t).
template <int DT, bool RELU_A>
__global__ __launch_bounds__(512, 2) void gemm_ph_kernel(const GemmParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 256, BN = 256, NT = 512, TM = 4, TN = 2, PL = 1;
  constexpr int SLABS = TM;
  constexpr int HALF = 128 * 128, TILE_BYTES = 4 * HALF;  // bytes: one half-tile, one k-tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 31, lh = lane >> 5;

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

  // loader: a wave-instruction moves 8 rows x 128 B; lane (r8 = lane>>3, kc = lane&7) owns chunk kc of row lrow (and of
  // row 64 + lrow) of every half-tile.  q = 2*h + z: half h, rows z*64 + lrow.
  const int kc = lane & 7, lrow = wave * 8 + (lane >> 3);
  const int sc = kc ^ ((lrow >> 1) & 7);
  int a_iy0[4], a_ix0[4];
  unsigned a_off[4], w_off[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int m = m0 + (q >> 1) * 128 + (q & 1) * 64 + lrow;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    int rem, ox;
    const int img = row_div(mm, p.a_rpi, p.a_rpi_rcp, false, rem);
    const int oy = row_div(rem, p.Wout, p.wout_rcp, false, ox);
    a_iy0[q] = ok ? oy * p.stride - p.pad_t : -0x40000000;
    a_ix0[q] = ox * p.stride - p.pad_l;
    const long long e = (long long)img * p.a_img_stride + p.a_off + ((long long)a_iy0[q] * p.Win + a_ix0[q]) * p.a_pix_stride + sc * 8;
    a_off[q] = (unsigned)(ok ? e * 2 : 0);
    w_off[q] = (unsigned)(((long long)(n0 + (q >> 1) * 128 + (q & 1) * 64 + lrow) * p.ldw + sc * 8) * 2);
  }
  const int w_bytes = (int)((long long)p.N * p.ldw * 2);
  const __amdgpu_buffer_rsrc_t rsrcA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.A), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.W), 0, w_bytes, 0x00020000);

  int ky = 0, kx = 0, c0 = 0;  // tap / channel offset of the k-tile being staged (wave-uniform)
  const int ld_base = wave * 8 * 128;
  // the two pieces of A half H (0 / 1) resp. W half H of the k-tile at (ky, kx, c0) into buffer BUF
#define DPTX_PH_PIECE_A(BUF, H, Z)                                                                                 \
  do {                                                                                                             \
    char* d_ = smem + (BUF) * TILE_BYTES + (H) * HALF + ld_base + (Z) * 64 * 128;                                  \
    const unsigned tap_ = (unsigned)(((ky * p.Win + kx) * p.a_pix_stride + c0) * 2);                               \
    const int iy = a_iy0[2 * (H) + (Z)] + ky, ix = a_ix0[2 * (H) + (Z)] + kx;                                      \
    const bool valid = ((unsigned)iy < (unsigned)p.Hin) && ((unsigned)ix < (unsigned)p.Win);                       \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcA, (__attribute__((address_space(3))) void*)d_, 16,               \
                                             valid ? a_off[2 * (H) + (Z)] + tap_ : OOB, 0, 0, 0);                  \
  } while (0)
#define DPTX_PH_PIECE_W(BUF, H, Z)                                                                                 \
  do {                                                                                                             \
    char* d_ = smem + (BUF) * TILE_BYTES + (2 + (H)) * HALF + ld_base + (Z) * 64 * 128;                            \
    const unsigned wk_ = (unsigned)((((ky * p.ksz + kx) * p.Cin) + c0) * 2);                                       \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void*)d_, 16,               \
                                             w_off[2 * (H) + (Z)] + wk_, 0, 0, 0);                                 \
  } while (0)
#define DPTX_PH_STAGE_A(BUF, H) do { DPTX_PH_PIECE_A(BUF, H, 0); DPTX_PH_PIECE_A(BUF, H, 1); } while (0)
#define DPTX_PH_STAGE_W(BUF, H) do { DPTX_PH_PIECE_W(BUF, H, 0); DPTX_PH_PIECE_W(BUF, H, 1); } while (0)
#define DPTX_PH_NEXT_TAP()                                                                                         \
  do {                                                                                                             \
    if (p.k_tap_fast) {                                                                                            \
      if (++kx == p.ksz) { kx = 0; if (++ky == p.ksz) { ky = 0; c0 += BK; } }                                      \
    } else {                                                                                                       \
      c0 += BK;                                                                                                    \
      if (c0 >= p.Cin) { c0 = 0; if (++kx == p.ksz) { kx = 0; ++ky; } }                                            \
    }                                                                                                              \
  } while (0)

  // fragment reads: row r of a half-tile is 128 B, chunk c at ((c ^ ((r >> 1) & 7)) << 4); the swizzle term of all of a
  // lane's rows is (lr >> 1) & 7 (the wave offsets are multiples of 32 rows)
  const int sw = (lr >> 1) & 7;
  const int a_rd = (wm * 64 + lr) * 128, b_rd = (wn * 32 + lr) * 128;
  int ch[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) ch[ks] = ((2 * ks + lh) ^ sw) << 4;

  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = p.K / BK;
  // prologue: all of tile 0
  DPTX_PH_STAGE_A(0, 0);
  DPTX_PH_STAGE_W(0, 0);
  DPTX_PH_STAGE_W(0, 1);
  DPTX_PH_STAGE_A(0, 1);
  DPTX_PH_NEXT_TAP();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (wm == 1) asm volatile("s_barrier" ::: "memory");  // group 1 runs one barrier behind

#define DPTX_PH_WAIT()                                                                                             \
  do {                                                                                                             \
    if (!stage) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
    else if (wm == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                             \
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                                          \
  } while (0)
#define DPTX_PH_MMA(I0, J, BF, HOOK0, HOOK1)                                                                       \
  do {                                                                                                             \
    DPTX_PH_PRIO(1);                                                                                               \
    acc[I0][J] = T16<DT>::mfma32(af[0][0], BF[0], acc[I0][J]);                                                     \
    acc[I0 + 1][J] = T16<DT>::mfma32(af[1][0], BF[0], acc[I0 + 1][J]);                                             \
    HOOK0;                                                                                                         \
    acc[I0][J] = T16<DT>::mfma32(af[0][1], BF[1], acc[I0][J]);                                                     \
    acc[I0 + 1][J] = T16<DT>::mfma32(af[1][1], BF[1], acc[I0 + 1][J]);                                             \
    acc[I0][J] = T16<DT>::mfma32(af[0][2], BF[2], acc[I0][J]);                                                     \
    acc[I0 + 1][J] = T16<DT>::mfma32(af[1][2], BF[2], acc[I0 + 1][J]);                                             \
    HOOK1;                                                                                                         \
    acc[I0][J] = T16<DT>::mfma32(af[0][3], BF[3], acc[I0][J]);                                                     \
    acc[I0 + 1][J] = T16<DT>::mfma32(af[1][3], BF[3], acc[I0 + 1][J]);                                             \
    DPTX_PH_PRIO(0);                                                                                               \
  } while (0)

#ifdef DPTX_PH_NOPRIO
#define DPTX_PH_PRIO(X) do { } while (0)
#else
#define DPTX_PH_PRIO(X) __builtin_amdgcn_s_setprio(X)
#endif
// DPTX_PH_SPLIT: 0 both DMA pieces of a phase in the load section, 1 one there and one between the MFMAs, 2 both
// between the MFMAs (the matrix pipe keeps executing while the wave is stuck in the DMA issue)
#ifndef DPTX_PH_SPLIT
#define DPTX_PH_SPLIT 0
#endif
#define DPTX_PH_L(KIND, BUF, H) do { if (stage) { if (DPTX_PH_SPLIT == 0) DPTX_PH_STAGE_##KIND(BUF, H); else if (DPTX_PH_SPLIT == 1) DPTX_PH_PIECE_##KIND(BUF, H, 0); } } while (0)
#define DPTX_PH_M0(KIND, BUF, H) do { if (stage && DPTX_PH_SPLIT == 2) DPTX_PH_PIECE_##KIND(BUF, H, 0); } while (0)
#define DPTX_PH_M1(KIND, BUF, H) do { if (stage && DPTX_PH_SPLIT >= 1) DPTX_PH_PIECE_##KIND(BUF, H, 1); } while (0)
#ifdef DPTX_PH_STAGE_FIRST   // experiment: DMA pieces in front of the fragment reads
#define DPTX_PH_LOADS(READS, STAGE) do { STAGE; READS; } while (0)
#else
#define DPTX_PH_LOADS(READS, STAGE) do { READS; STAGE; } while (0)
#endif
  for (int kt = 0; kt < nk; ++kt) {
    const bool stage = kt + 1 < nk;
    const int cb = kt & 1, nb = cb ^ 1;
    const char* t_ = smem + cb * TILE_BYTES;
    u32x4_t af[2][4], b0[4], b1[4];
    // ---- phase 0: (top, j0)
    DPTX_PH_LOADS(
        {
          _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) b0[ks] = *(const u32x4_t*)(t_ + 2 * HALF + b_rd + ch[ks]);
          _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {
            af[0][ks] = *(const u32x4_t*)(t_ + a_rd + ch[ks]);
            af[1][ks] = *(const u32x4_t*)(t_ + a_rd + 32 * 128 + ch[ks]);
          }
        },
        { DPTX_PH_L(A, nb, 0); });
    asm volatile("s_barrier" ::: "memory");
    if (RELU_A) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { af[0][ks] = relu8(af[0][ks]); af[1][ks] = relu8(af[1][ks]); }
    }
    DPTX_PH_MMA(0, 0, b0, DPTX_PH_M0(A, nb, 0), DPTX_PH_M1(A, nb, 0));
    DPTX_PH_WAIT();
    asm volatile("s_barrier" ::: "memory");
    // ---- phase 1: (top, j1)
    DPTX_PH_LOADS({ _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) b1[ks] = *(const u32x4_t*)(t_ + 3 * HALF + b_rd + ch[ks]); },
                  { DPTX_PH_L(W, nb, 0); });
    asm volatile("s_barrier" ::: "memory");
    DPTX_PH_MMA(0, 1, b1, DPTX_PH_M0(W, nb, 0), DPTX_PH_M1(W, nb, 0));
    DPTX_PH_WAIT();
    asm volatile("s_barrier" ::: "memory");
    // ---- phase 2: (bottom, j1)
    DPTX_PH_LOADS(
        {
          _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {
            af[0][ks] = *(const u32x4_t*)(t_ + HALF + a_rd + ch[ks]);
            af[1][ks] = *(const u32x4_t*)(t_ + HALF + a_rd + 32 * 128 + ch[ks]);
          }
        },
        { DPTX_PH_L(W, nb, 1); });
    asm volatile("s_barrier" ::: "memory");
    if (RELU_A) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) { af[0][ks] = relu8(af[0][ks]); af[1][ks] = relu8(af[1][ks]); }
    }
    DPTX_PH_MMA(2, 1, b1, DPTX_PH_M0(W, nb, 1), DPTX_PH_M1(W, nb, 1));
    DPTX_PH_WAIT();
    asm volatile("s_barrier" ::: "memory");
    // ---- phase 3: (bottom, j0)
    DPTX_PH_LOADS({ _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) b0[ks] = *(const u32x4_t*)(t_ + 2 * HALF + b_rd + ch[ks]); },
                  { DPTX_PH_L(A, nb, 1); });
    asm volatile("s_barrier" ::: "memory");
    DPTX_PH_MMA(2, 0, b0, DPTX_PH_M0(A, nb, 1), DPTX_PH_M1(A, nb, 1));
    if (stage) DPTX_PH_NEXT_TAP();
    DPTX_PH_WAIT();
    asm volatile("s_barrier" ::: "memory");
  }
  if (wm == 0) asm volatile("s_barrier" ::: "memory");  // group 0 catches up
#undef DPTX_PH_MMA
#undef DPTX_PH_WAIT
#undef DPTX_PH_LOADS
#undef DPTX_PH_PRIO
#undef DPTX_PH_NEXT_TAP
#undef DPTX_PH_STAGE_W
#undef DPTX_PH_STAGE_A
#undef DPTX_PH_PIECE_W
#undef DPTX_PH_PIECE_A
#undef DPTX_PH_L
#undef DPTX_PH_M0
#undef DPTX_PH_M1
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  epilogue<DT, BM, BN, TM, TN, PL, NT, SLABS, true>(p, smem, m0, n0, wm, wn, lr, lh, tid, acc);
#endif
}
