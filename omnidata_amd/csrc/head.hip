// head.hip -- the tail of the DPT head in ONE kernel (dpt_depth.py:93-98):
//   Interpolate(x2, bilinear, align_corners=True) -> Conv2d(128, 32, 3, pad 1) -> ReLU -> Conv2d(32, C, 1) -> ReLU
//   H0 NHWC 16-bit [B, Hs, Ws, 128]  ->  y NCHW fp32 [B, C, 2Hs, 2Ws]
//
// Unfused, this is the most HBM-hungry stretch of the forward: the up-sampled 128-channel map is 37.7 MB per
// image (written once, read through nine conv taps), the 32-channel map another 9.4 MB -- 1.25 ms of 16.1 ms at
// B=32.  Here neither exists in memory.  A block owns 8x32 output pixels at a time:
//   * its 10x34-pixel window of the UP-SAMPLED map is built in LDS (87 KB) straight from the 6x18 source pixels
//     it depends on: a thread owns one (window column, 8-channel chunk), keeps the 6x2 source vectors it needs in
//     registers (fetched one tile ahead, so the global latency hides behind the previous tile's MFMAs) and emits
//     the 10 window rows with the same fp32 formula and the same 16-bit rounding as upsample2x_kernel;
//   * the whole 3x3 weight tensor [32][9][128] (73.7 KB) lives in LDS for the life of the (persistent) block;
//   * the conv is 72 v_mfma_f32_32x32x16 per wave (wave w = output row w) with both operands read from LDS
//     (16-B chunks XOR-swizzled: conflict-free ds_read_b128), computed TRANSPOSED (rows = the 32 output channels,
//     columns = the 32 pixels) so that a lane ends up with 16 channels of ONE pixel: bias + ReLU + the 1x1
//     projection to C channels are a 16-term dot product per lane plus one cross-half shuffle, and the result is
//     stored as 128-B rows of the NCHW fp32 output.
// k order = (tap, channel), the implicit-GEMM's: the fp32 accumulation sequence is the unfused path's.
// 576 threads: waves 0-7 own window columns 0..31 and the MFMA rows, wave 8 owns the two halo columns.
// LDS 161.3 KB -> one block per CU; grid = min(tiles, CUs), each block walks a contiguous run of tiles.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace dptx {

constexpr int HT_THREADS = 576;
constexpr int HT_PR = 10, HT_PC = 34;                 // window rows / columns (8x32 outputs + 1-pixel halo)
constexpr int HT_W_BYTES = 32 * 9 * 256;              // [n][tap][16 chunks x 16 B]
constexpr int HT_P_BYTES = HT_PR * HT_PC * 256;       // [row][col][16 chunks x 16 B]
constexpr int HT_CONST_FLOATS = 32 + 3 * 32 + 4;      // bias2, w4 (<= 3 channels), bias4
constexpr size_t HT_SMEM = HT_W_BYTES + HT_P_BYTES + HT_CONST_FLOATS * 4;

template <int DT>
__device__ __forceinline__ void unpack8(const u32x4_t v, float* f) {
  f[0] = T16<DT>::tof((uint16_t)(v.x & 0xffffu)); f[1] = T16<DT>::tof((uint16_t)(v.x >> 16));
  f[2] = T16<DT>::tof((uint16_t)(v.y & 0xffffu)); f[3] = T16<DT>::tof((uint16_t)(v.y >> 16));
  f[4] = T16<DT>::tof((uint16_t)(v.z & 0xffffu)); f[5] = T16<DT>::tof((uint16_t)(v.z >> 16));
  f[6] = T16<DT>::tof((uint16_t)(v.w & 0xffffu)); f[7] = T16<DT>::tof((uint16_t)(v.w >> 16));
}

// ATen's bilinear association ly0*(lx0*a + lx1*b) + ly1*(lx0*c + lx1*d) is separable as written: the inner sums are the
// horizontal blends of source rows y0 and y1.  They are formed once per source row (hblend), and every window row is
// one vertical blend of two of them (vblend) -- bit-identical to bilerp() in common.h, 2.3x fewer VALU operations.
template <int DT>
__device__ __forceinline__ void hblend(const u32x4_t s0, const u32x4_t s1, float lx0, float lx1, float (&t)[8]) {
  float a[8], b[8];
  unpack8<DT>(s0, a);
  unpack8<DT>(s1, b);
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = __fmaf_rn(lx1, b[e], __fmul_rn(lx0, a[e]));
}
template <int DT, int J>
__device__ __forceinline__ u32x4_t vblend(const float (&T)[6][8], float ly0, float ly1) {
  float o[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = __fmaf_rn(ly1, T[J + 1][e], __fmul_rn(ly0, T[J][e]));
  u32x4_t r;
  r.x = T16<DT>::pack2(o[0], o[1]);
  r.y = T16<DT>::pack2(o[2], o[3]);
  r.z = T16<DT>::pack2(o[4], o[5]);
  r.w = T16<DT>::pack2(o[6], o[7]);
  return r;
}

template <int DT>
__global__ __launch_bounds__(HT_THREADS) void head_tail_kernel(const uint16_t* __restrict__ H0, const uint16_t* __restrict__ W2,
                                                               const float* __restrict__ b2, const float* __restrict__ w4,
                                                               const float* __restrict__ b4, void* __restrict__ y, int io, int B,
                                                               int Hs, int Ws, int C, int relu_out, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Wl = smem;
  char* P = smem + HT_W_BYTES;
  float* cst = (float*)(smem + HT_W_BYTES + HT_P_BYTES);  // [0,32) bias2, [32, 32+32C) w4, [128, 128+C) bias4

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int Ho = 2 * Hs, Wo = 2 * Ws;
  const int tiles_x = Wo / 32, tiles_y = Ho / 8, tpi = tiles_x * tiles_y;
  const float ry = Ho > 1 ? (float)(Hs - 1) / (float)(Ho - 1) : 0.f;
  const float rx = Wo > 1 ? (float)(Ws - 1) / (float)(Wo - 1) : 0.f;

  // ---- once per block: weights and constants -> LDS
  {
    constexpr int NW = 32 * 9 * 16 / HT_THREADS;  // = 8 vectors of 16 B per thread, all in flight before the first write
    static_assert(NW * HT_THREADS == 32 * 9 * 16, "weight tensor splits evenly over the block");
    u32x4_t wv[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int i = tid + j * HT_THREADS;
      const int n = i / 144, rem = i - n * 144;
      wv[j] = *(const u32x4_t*)(W2 + n * 1152 + (rem >> 4) * 128 + (rem & 15) * 8);
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int i = tid + j * HT_THREADS;
      const int n = i / 144, rem = i - n * 144, tap = rem >> 4, ch = rem & 15;
      *(u32x4_t*)(Wl + n * 2304 + tap * 256 + ((ch ^ (n & 15)) << 4)) = wv[j];
    }
  }
  if (tid < 32) cst[tid] = b2[tid];
  if (tid < 32 * C) cst[32 + tid] = w4[tid];
  if (tid < C) cst[128 + tid] = b4[tid];

  const int per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int t_beg = blockIdx.x * per, t_end = min(t_beg + per, ntiles);

  // window column owned by this thread
  const int wx = tid >> 4, wch = tid & 15;
  const bool col_thread = wx < HT_PC;

  u32x4_t S[6][2];
  const u32x4_t zero4 = {0u, 0u, 0u, 0u};

  auto fetch = [&](int t) {  // source vectors of tile t for this thread's column
    const int b = t / tpi, rem = t - b * tpi, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int ox = tx * 32 - 1 + wx;
    const bool vx = col_thread && ox >= 0 && ox < Wo;
    const float sx = rx * (float)(vx ? ox : 0);
    const int x0 = (int)sx, x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
    const int oyb = max(ty * 8 - 1, 0);
    const int ybase = (int)(ry * (float)oyb);
    const uint16_t* img = H0 + (long long)b * Hs * Ws * 128 + wch * 8;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int row = min(ybase + j, Hs - 1);
      S[j][0] = vx ? *(const u32x4_t*)(img + ((long long)row * Ws + x0) * 128) : zero4;
      S[j][1] = vx ? *(const u32x4_t*)(img + ((long long)row * Ws + x1) * 128) : zero4;
    }
  };

  if (t_beg < t_end) fetch(t_beg);
  __syncthreads();

  for (int t = t_beg; t < t_end; ++t) {
    const int b = t / tpi, rem = t - b * tpi, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int oy0 = ty * 8, ox0 = tx * 32;

    // ---- up-sampled window -> LDS (zero outside the image: the conv's padding)
    if (col_thread) {
      const int ox = ox0 - 1 + wx;
      const bool vx = ox >= 0 && ox < Wo;
      const float sx = rx * (float)(vx ? ox : 0);
      const int x0 = (int)sx;
      const float lx1 = sx - (float)x0, lx0 = 1.f - lx1;
      const int ybase = (int)(ry * (float)max(oy0 - 1, 0));
      char* dst = P + wx * 256 + ((wch ^ (wx & 15)) << 4);
      float T[6][8];
#pragma unroll
      for (int j = 0; j < 6; ++j) hblend<DT>(S[j][0], S[j][1], lx0, lx1, T[j]);
      for (int r = 0; r < HT_PR; ++r) {
        const int oy = oy0 - 1 + r;
        u32x4_t o = zero4;
        if (vx && oy >= 0 && oy < Ho) {
          const float sy = ry * (float)oy;
          const int y0 = (int)sy;
          const float ly1 = sy - (float)y0, ly0 = 1.f - ly1;
          switch (y0 - ybase) {  // uniform over the block; slot j+1 holds row min(y0+1, Hs-1) = ATen's y1
            case 0: o = vblend<DT, 0>(T, ly0, ly1); break;
            case 1: o = vblend<DT, 1>(T, ly0, ly1); break;
            case 2: o = vblend<DT, 2>(T, ly0, ly1); break;
            case 3: o = vblend<DT, 3>(T, ly0, ly1); break;
            default: o = vblend<DT, 4>(T, ly0, ly1); break;
          }
        }
        *(u32x4_t*)(dst + r * (HT_PC * 256)) = o;
      }
    }
    if (t + 1 < t_end) fetch(t + 1);  // in flight across the barrier and the MFMA phase
    __syncthreads();

    // ---- 3x3 conv 128 -> 32 on the window, transposed: acc[r] = channel (r&3)+8(r>>2)+4*lh of pixel lr
    if (wave < 8) {
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const char* wrow = Wl + lr * 2304;
      const int wkey = lr & 15;
      // 72 k-steps in 36 batches of 2; the fragment reads of batch i+1 are issued before the MFMAs of batch i
      // (two register sets; 4 would not fit beside the 48 prefetch registers at 3 waves per SIMD)
      constexpr int NB = 2, NBATCH = 8 / NB * 9;
      u32x4_t wf[2][NB], pf[2][NB];
      auto read_batch = [&](int bi, u32x4_t (&w)[NB], u32x4_t (&q)[NB]) {
        const int tap = bi / (8 / NB), cb0 = (bi % (8 / NB)) * NB;
        const int ky = tap / 3, kx = tap - ky * 3;
        const int xc = lr + kx;
        const char* prow = P + ((wave + ky) * HT_PC + xc) * 256;
        const int pkey = xc & 15;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
          const int chunk = 2 * (cb0 + i) + lh;
          w[i] = *(const u32x4_t*)(wrow + tap * 256 + ((chunk ^ wkey) << 4));
          q[i] = *(const u32x4_t*)(prow + ((chunk ^ pkey) << 4));
        }
      };
      read_batch(0, wf[0], pf[0]);
#pragma unroll
      for (int bi = 0; bi < NBATCH; ++bi) {
        if (bi + 1 < NBATCH) read_batch(bi + 1, wf[(bi + 1) & 1], pf[(bi + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);  // keep the reads of batch bi+1 ahead of the MFMAs of batch bi
#pragma unroll
        for (int i = 0; i < NB; ++i) acc = T16<DT>::mfma32(wf[bi & 1][i], pf[bi & 1][i], acc);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- bias + ReLU, 1x1 conv to C channels, final ReLU, NCHW fp32 rows
      float h[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bb = *(const float4*)(cst + 8 * q + 4 * lh);
        h[4 * q + 0] = fmaxf(acc[4 * q + 0] + bb.x, 0.f);
        h[4 * q + 1] = fmaxf(acc[4 * q + 1] + bb.y, 0.f);
        h[4 * q + 2] = fmaxf(acc[4 * q + 2] + bb.z, 0.f);
        h[4 * q + 3] = fmaxf(acc[4 * q + 3] + bb.w, 0.f);
      }
      const int oy = oy0 + wave;
      for (int c = 0; c < C; ++c) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 ww = *(const float4*)(cst + 32 + c * 32 + 8 * q + 4 * lh);
          s += ww.x * h[4 * q + 0] + ww.y * h[4 * q + 1] + ww.z * h[4 * q + 2] + ww.w * h[4 * q + 3];
        }
        s += __shfl_xor(s, 32);
        s += cst[128 + c];
        if (relu_out) s = fmaxf(s, 0.f);
        if ((c & 1) == lh) io_store(y, (((long long)b * C + c) * Ho + oy) * Wo + ox0 + lr, s, io);
      }
    }
    __syncthreads();  // the window is rebuilt for the next tile
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same tail for the hi/lo-plane modes (fp16x3 and the head of the parity mode "mixed"): H0 and W2 are plane pairs
// (lo = x - hi at a fixed distance) and every product is three MFMAs (hi*lo + lo*hi + hi*hi).  Unfused this stretch is 3 ms of
// the 20.8 ms parity-mode forward at B = 32 (head.up moves four 37.7 MB/image planes, the 3-MFMA conv with N = 32 runs at
// 520 TFLOP/s-equivalent, head.out re-reads two planes): profiles/r04_experiments.md.
//
// Two planes of the 10 x 34 window (174 KB) and of the weights (147 KB) do not fit 160 KB of LDS, so the K axis is split into
// four CHANNEL QUARTERS of 32 that are processed one after the other into the same accumulators.  Per quarter a block holds
//   W  [plane][tap][n = 32][32 ch]        = 36 864 B  (LDS-DMA from L2 per tile and quarter: lane-linear 16-row pieces, XOR
//                                                      swizzle on the SOURCE chunk -- conflict-free ds_read_b128 of 64-byte rows)
//   P  [plane][10 rows][34 cols][32 ch]   = 43 520 B  (a thread builds five window rows of one column and eight channels from
//                                                      4 x 2 x 2 source vectors)
// = 78.5 KB per stage, TWO stages (157 KB): the block is warp-specialised -- waves 0..4 build the window and wave 5 streams
// the weights of step g + 1 while waves 6..9 (the multipliers, two output rows each: the weight fragments are read once for
// both) run the 18 k-steps x 2 rows x 3 MFMAs of step g; one s_barrier per step.  The first forms of this kernel ran the
// two phases one after the other in all waves (one 161 KB block with channel halves, then two 78.5 KB blocks per CU): 1.53 -
// 1.67 ms per launch at B = 32 -- the phases of co-resident blocks did not overlap by themselves.  This form: 1.43 ms;
// with the MFMA phase removed 0.95 ms, with the window build removed 0.94 ms, without the weight DMA 1.38 ms
// (tools/gpu/r4_headx3_bench.py, DPTX_HX_DBG).  Both halves sit at twice their own floor: the multipliers read one 1 KB
// fragment from LDS per MFMA (w_hi, w_lo, and hi / lo of two rows for six MFMAs), which is the LDS port's rate, and the
// builders' ds_write_b128 share that port -- a deeper register blocking (four rows per wave) needs a 16-row window that does
// not fit two stages.  The weights are re-streamed per tile (147 KB per 8 x 32 pixels, ~21 MB per CU and forward out of L2).
constexpr int HX_THREADS = 640;                 // waves 0..4 build the window, wave 5 streams the weights, waves 6..9 multiply
constexpr int HX_WPL = 9 * 32 * 64;             // bytes of one weight plane of one channel quarter: [tap][n][4 chunks x 16 B]
constexpr int HX_PPL = HT_PR * HT_PC * 64;      // bytes of one window plane of one channel quarter
constexpr int HX_STAGE = 2 * HX_WPL + 2 * HX_PPL;   // 80 384 B: [W hi | W lo | P hi | P lo]
constexpr size_t HX_SMEM = 2 * HX_STAGE;        // two stages: 160 768 B (the head constants stay in memory)
static_assert(HX_SMEM <= 160 * 1024, "one block per CU");

template <int DT>
__device__ __forceinline__ void hblend2(const u32x4_t (&s0)[2], const u32x4_t (&s1)[2], float lx0, float lx1, float (&t)[8]) {
  float a[8], b[8];
  unpack8x<DT, 2>(s0[0], s0[1], a);   // hi + lo
  unpack8x<DT, 2>(s1[0], s1[1], b);
#pragma unroll
  for (int e = 0; e < 8; ++e) t[e] = __fmaf_rn(lx1, b[e], __fmul_rn(lx0, a[e]));
}

template <int DT>
__global__ __launch_bounds__(HX_THREADS) void head_tail_x3_kernel(const uint16_t* __restrict__ H0, const uint16_t* __restrict__ W2,
                                                                  const float* __restrict__ b2, const float* __restrict__ w4,
                                                                  const float* __restrict__ b4, void* __restrict__ y, int io, int B,
                                                                  int Hs, int Ws, int C, int relu_out, int ntiles, long long plane,
                                                                  long long wplane, int dbg) {
  // dbg (DPTX_HX_DBG, timing ablations only -- results are wrong): 1 no MFMA phase, 2 no window build, 4 no weight DMA
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int Ho = 2 * Hs, Wo = 2 * Ws;
  const int tiles_x = Wo / 32, tiles_y = Ho / 8, tpi = tiles_x * tiles_y;
  const float ry = Ho > 1 ? (float)(Hs - 1) / (float)(Ho - 1) : 0.f;
  const float rx = Wo > 1 ? (float)(Ws - 1) / (float)(Wo - 1) : 0.f;

  const int per = (ntiles + gridDim.x - 1) / gridDim.x;
  const int t_beg = blockIdx.x * per, t_end = min(t_beg + per, ntiles);
  const int G = (t_end - t_beg) * 4;   // steps of this block: (tile, channel quarter); step g lives in LDS stage g & 1
  if (G <= 0) return;

  // Schedule (one s_barrier per step, G of them on either side):
  //   builders:    for g: build(g) into stage g & 1;  barrier B_g
  //   multipliers: barrier B_0;  for g: consume(g) out of stage g & 1;  barrier B_(g+1) unless g is the last step
  // Before B_g the builders write stage g & 1 while the multipliers read stage (g - 1) & 1; behind B_g the builders move on to
  // stage (g + 1) & 1 = the one the multipliers have just finished with.
  if (wave == 5) {
    // ============================================================ weight wave: 36 LDS-DMA pieces per step
    // piece q (0..35) = [plane = q / 18][rows 16 (q % 18) .. + 15 of (tap, n)][4 chunks]; lane l -> row r = 16 (q % 18) + l / 4
    // (tap = r / 32, n = r % 32), LDS chunk l % 4 holds SOURCE chunk (l % 4) ^ ((n >> 2) & 3)
    const __amdgpu_buffer_rsrc_t rsrcW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(W2), 0, (int)((wplane + 32 * 1152) * 2), 0x00020000);
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
      const int quarter = g & 3;
      char* Wl = smem + (g & 1) * HX_STAGE;
      if (!(dbg & 4)) {
#pragma unroll 6
        for (int q = 0; q < 36; ++q) {
          const int pl = q / 18, r = 16 * (q % 18) + (lane >> 2);
          const int tap = r >> 5, n = r & 31;
          const int sch = (lane & 3) ^ ((n >> 2) & 3);
          const unsigned off = (unsigned)(((long long)pl * wplane + n * 1152 + tap * 128 + quarter * 32 + sch * 8) * 2);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcW, (__attribute__((address_space(3))) void*)(Wl + q * 1024), 16, off, 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");   // B_g
    }
    return;
  }
  if (wave < 5) {
    // ============================================================ window builders (320 threads, 272 items)
    // an ITEM is (window column 0..33, 8-channel chunk 0..3 of the quarter, row half): five window rows of one column from four
    // source rows
    const int item = tid;
    const int rh = item >= 136 ? 1 : 0, irem = item - 136 * rh, wx = irem >> 2, wch = irem & 3;
    typedef u32x4_t SrcVecs[4][2][2];  // [source row][x0 / x1][plane]
    const u32x4_t zero4 = {0u, 0u, 0u, 0u};
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
      const int t = t_beg + (g >> 2), quarter = g & 3;
      char* P = smem + (g & 1) * HX_STAGE + 2 * HX_WPL;
      const int b = t / tpi, rem = t - b * tpi, ty = rem / tiles_x, tx = rem - ty * tiles_x;
      const int oy0 = ty * 8, ox0 = tx * 32;
      if (item < 2 * 136 && !(dbg & 2)) {
        const int ox = ox0 - 1 + wx;
        const bool vx = ox >= 0 && ox < Wo;
        const float sx = rx * (float)(vx ? ox : 0);
        const int x0 = (int)sx, x1 = x0 + (x0 < Ws - 1 ? 1 : 0);
        const float lx1 = sx - (float)x0, lx0 = 1.f - lx1;
        const int ybase = (int)(ry * (float)max(oy0 - 1 + 5 * rh, 0));
        const uint16_t* img = H0 + (long long)b * Hs * Ws * 128 + quarter * 32 + wch * 8;
        SrcVecs S;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = min(ybase + j, Hs - 1);
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            S[j][0][pl] = vx ? *(const u32x4_t*)(img + pl * plane + ((long long)row * Ws + x0) * 128) : zero4;
            S[j][1][pl] = vx ? *(const u32x4_t*)(img + pl * plane + ((long long)row * Ws + x1) * 128) : zero4;
          }
        }
        float T[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j) hblend2<DT>(S[j][0], S[j][1], lx0, lx1, T[j]);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
          const int r = 5 * rh + k, oy = oy0 - 1 + r;
          float o[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = 0.f;
          if (vx && oy >= 0 && oy < Ho) {
            const float sy = ry * (float)oy;
            const int y0 = (int)sy;
            const float ly1 = sy - (float)y0, ly0 = 1.f - ly1;
            const int j0 = y0 - ybase;  // 0..2; slot j0 + 1 holds row min(y0 + 1, Hs - 1) = ATen's y1
            if (j0 == 0) {
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = __fmaf_rn(ly1, T[1][e], __fmul_rn(ly0, T[0][e]));
            } else if (j0 == 1) {
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = __fmaf_rn(ly1, T[2][e], __fmul_rn(ly0, T[1][e]));
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = __fmaf_rn(ly1, T[3][e], __fmul_rn(ly0, T[2][e]));
            }
          }
          // hi / lo split exactly as store8f<DT, 2> (zero outside the image: the conv's padding)
          const uint4 hi = pack8<DT>(o);
          float hf[8], lf[8];
          unpack8<DT>(hi, hf);
#pragma unroll
          for (int e = 0; e < 8; ++e) lf[e] = o[e] - hf[e];
          const uint4 lo = pack8<DT>(lf);
          const int idx = r * HT_PC + wx;
          char* dst = P + idx * 64 + ((wch ^ ((idx >> 2) & 3)) << 4);
          *(u32x4_t*)dst = u32x4_t{hi.x, hi.y, hi.z, hi.w};
          *(u32x4_t*)(dst + HX_PPL) = u32x4_t{lo.x, lo.y, lo.z, lo.w};
        }
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's window stores have landed
      asm volatile("s_barrier" ::: "memory");                        // B_g
    }
    return;
  }

  // ============================================================== multipliers (waves 6..9): output rows 2 mw, 2 mw + 1
  const int mw = wave - 6;
  // fragment addresses inside a stage (bytes): chunk = 2 cb + lh sits at position chunk ^ key, so cb = 1 is the cb = 0 address
  // with bit 5 flipped.  W: row (tap, n = lr) -> the tap is an immediate; P: one base per (tap, output row)
  const int w_a0 = lr * 64 + ((lh ^ ((lr >> 2) & 3)) << 4);
  int p_a0[9][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = (2 * mw + i + tap / 3) * HT_PC + lr + tap % 3;
      p_a0[tap][i] = 2 * HX_WPL + idx * 64 + ((lh ^ ((idx >> 2) & 3)) << 4);
    }
  f32x16_t acc[2];
  asm volatile("s_barrier" ::: "memory");   // B_0
#pragma unroll 1
  for (int g = 0; g < G; ++g) {
    const int t = t_beg + (g >> 2), quarter = g & 3;
    const char* St = smem + (g & 1) * HX_STAGE;
    if (quarter == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
    }
    // 18 k-steps x 2 rows x (hi*lo + lo*hi + hi*hi); transposed: acc[i][r] = channel (r&3)+8(r>>2)+4 lh of pixel lr of row 2 mw + i
    u32x4_t wf[2][2], pf[2][2][2];  // [set][plane], [set][row][plane]: the reads run one k-step ahead of the MFMAs (two ahead
                                    // and s_setprio on these waves measured the same: profiles/r04_experiments.md)
    auto read_ks = [&](int tap, int cb, u32x4_t (&w)[2], u32x4_t (&q)[2][2]) {   // tap, cb: compile-time after unrolling
      const char* wr = St + ((w_a0 ^ (cb << 5)) + tap * 2048);
      w[0] = *(const u32x4_t*)wr;
      w[1] = *(const u32x4_t*)(wr + HX_WPL);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const char* pr = St + (p_a0[tap][i] ^ (cb << 5));
        q[i][0] = *(const u32x4_t*)pr;
        q[i][1] = *(const u32x4_t*)(pr + HX_PPL);
      }
    };
    if (!(dbg & 1)) {
      read_ks(0, 0, wf[0], pf[0]);
#pragma unroll
      for (int ks = 0; ks < 18; ++ks) {
        if (ks + 1 < 18) read_ks((ks + 1) >> 1, (ks + 1) & 1, wf[(ks + 1) & 1], pf[(ks + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        // (the two rows alternate: consecutive MFMAs never share an accumulator)
        acc[0] = T16<DT>::mfma32(wf[ks & 1][0], pf[ks & 1][0][1], acc[0]);   // w_hi * a_lo
        acc[1] = T16<DT>::mfma32(wf[ks & 1][0], pf[ks & 1][1][1], acc[1]);
        acc[0] = T16<DT>::mfma32(wf[ks & 1][1], pf[ks & 1][0][0], acc[0]);   // w_lo * a_hi
        acc[1] = T16<DT>::mfma32(wf[ks & 1][1], pf[ks & 1][1][0], acc[1]);
        acc[0] = T16<DT>::mfma32(wf[ks & 1][0], pf[ks & 1][0][0], acc[0]);   // w_hi * a_hi
        acc[1] = T16<DT>::mfma32(wf[ks & 1][0], pf[ks & 1][1][0], acc[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (quarter == 3) {
      const int b = t / tpi, rem = t - b * tpi, ty = rem / tiles_x, tx = rem - ty * tiles_x;
      const int oy0 = ty * 8, ox0 = tx * 32;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float h[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 bb = *(const float4*)(b2 + 8 * q + 4 * lh);
          h[4 * q + 0] = fmaxf(acc[i][4 * q + 0] + bb.x, 0.f);
          h[4 * q + 1] = fmaxf(acc[i][4 * q + 1] + bb.y, 0.f);
          h[4 * q + 2] = fmaxf(acc[i][4 * q + 2] + bb.z, 0.f);
          h[4 * q + 3] = fmaxf(acc[i][4 * q + 3] + bb.w, 0.f);
        }
        const int oy = oy0 + 2 * mw + i;
        for (int c = 0; c < C; ++c) {
          float sacc = 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 ww = *(const float4*)(w4 + c * 32 + 8 * q + 4 * lh);
            sacc += ww.x * h[4 * q + 0] + ww.y * h[4 * q + 1] + ww.z * h[4 * q + 2] + ww.w * h[4 * q + 3];
          }
          sacc += __shfl_xor(sacc, 32);
          sacc += b4[c];
          if (relu_out) sacc = fmaxf(sacc, 0.f);
          if ((c & 1) == lh) io_store(y, (((long long)b * C + c) * Ho + oy) * Wo + ox0 + lr, sacc, io);
        }
      }
    }
    if (g + 1 < G) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragment reads of stage g & 1 are done
      asm volatile("s_barrier" ::: "memory");               // B_(g+1)
    }
  }
#endif
}

hipError_t launch_head_tail(int mode, const void* H0, const void* W2, const float* b2, const float* w4, const float* b4,
                            void* y, int io, int B, int Hs, int Ws, int C, int relu_out, hipStream_t stream, Planes pl) {
  if (C < 1 || C > 3 || (2 * Hs) % 8 != 0 || (2 * Ws) % 32 != 0) return hipErrorInvalidValue;
  const int ntiles = B * ((2 * Hs) / 8) * ((2 * Ws) / 32);
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  const int grid = ntiles < cus ? ntiles : cus;
  if (mode_is_x3(mode)) {   // hi / lo planes, three MFMAs per product (head_tail_x3_kernel)
    // fp16 planes only (the parity mode "mixed" and fp16x3): the bf16 conversions of the window builder need more registers
    // than two waves per SIMD have (92 bytes of scratch); bf16x3 -- the range fallback -- keeps the three-launch tail
    if (mode != MODE_FP16X3 || pl.act == 0 || pl.w == 0 || (pl.w + 32 * 1152) * 2 >= (1ll << 31)) return hipErrorInvalidValue;
    auto k = head_tail_x3_kernel<DT_FP16>;
    ensure_dyn_smem((const void*)k, HX_SMEM);
    const int grid2 = grid;   // one block per CU
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("DPTX_HX_DBG"); dbg = e ? atoi(e) : 0; }
    hipLaunchKernelGGL(k, dim3(grid2), dim3(HX_THREADS), HX_SMEM, stream, (const uint16_t*)H0, (const uint16_t*)W2, b2, w4, b4, y, io, B,
                       Hs, Ws, C, relu_out, ntiles, pl.act, pl.w, dbg);
    return hipGetLastError();
  }
  if (mode == MODE_BF16) {
    auto k = head_tail_kernel<DT_BF16>;
    ensure_dyn_smem((const void*)k, HT_SMEM);
    hipLaunchKernelGGL(k, dim3(grid), dim3(HT_THREADS), HT_SMEM, stream, (const uint16_t*)H0, (const uint16_t*)W2, b2, w4, b4, y, io, B,
                       Hs, Ws, C, relu_out, ntiles);
  } else {
    auto k = head_tail_kernel<DT_FP16>;
    ensure_dyn_smem((const void*)k, HT_SMEM);
    hipLaunchKernelGGL(k, dim3(grid), dim3(HT_THREADS), HT_SMEM, stream, (const uint16_t*)H0, (const uint16_t*)W2, b2, w4, b4, y, io, B,
                       Hs, Ws, C, relu_out, ntiles);
  }
  return hipGetLastError();
}

}  // namespace dptx
