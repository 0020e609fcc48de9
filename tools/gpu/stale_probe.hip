// stale_probe.hip -- does a consumer kernel ever observe STALE data of the producer kernel launched just before it on
// the SAME stream while a second stream keeps the GPU busy?   (diagnostic tool, not part of the product)
//
// Round 1 removed a single-launch GroupNorm because "reading the output of the GEMM launched just before it on the same
// stream, it returned stale values in ~35 % of forwards whenever a second stream kept the GPU busy" and the kernel itself
// was never committed.  This probe rebuilds the situation from its parts and measures each producer->consumer edge type
// the engine has, tens of thousands of times, with the second stream saturating the chip:
//
//   stream A:  W(i): 128x128-tile "epilogue" stores of a pattern f(i, addr) into X[img][576][1024] (16-bit), with a
//                    randomly delayed tail block (the GEMM's last tiles finish late when CUs are taken by stream B)
//              R(i): block (img, 64-channel slab) reads its 576 x 128-B strided rows right away and compares with f(i, .)
//                    variants: plain global loads | buffer_load ... lds (LDS-DMA, what the GEMM's A path uses) |
//                    in-place read-modify-write (the removed kernel wrote its result over its input)
//   stream B:  back-to-back streaming kernels (HBM + LDS traffic) sized to overlap everything on stream A
//
// It also records device timestamps: max over W's blocks of their end time and min over R's blocks of their start time
// per iteration; W_end > R_start would mean the two launches overlapped (an ordering failure rather than a cache one).
//
// Build / run (on the GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/stale_probe tools/gpu/stale_probe.hip &&
//                                /tmp/stale_probe [iters=20000]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

constexpr int IMGS = 16, PIX = 576, CH = 1024;          // X[16][576][1024] u16 = 18.9 MB (a 1/16-resolution stage-2 map)
constexpr int ROWS = IMGS * PIX;

__device__ __forceinline__ unsigned short pat(unsigned iter, unsigned row, unsigned col) {
  unsigned h = iter * 2654435761u ^ (row * 40503u + col * 9973u);
  h ^= h >> 15;
  return (unsigned short)(h & 0xffffu);
}

struct Times { unsigned long long w_end_max, r_start_min; };

// W: grid = (ROWS/128) x (CH/128) tiles; thread t writes 16-B pieces of rows t/16 + 16*j
__global__ __launch_bounds__(256) void writer(unsigned short* X, unsigned iter, Times* tm, unsigned delay_tile, int delay_cycles) {
  const int tile = blockIdx.x;
  const int tm_ = tile / (CH / 128), tn = tile % (CH / 128);
  if ((unsigned)tile == delay_tile) {  // a late tail tile
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)delay_cycles) {}
  }
  const int c8 = threadIdx.x & 15, r0 = threadIdx.x >> 4;
  for (int j = 0; j < 8; ++j) {
    const int row = tm_ * 128 + r0 + 16 * j, col = tn * 128 + c8 * 8;
    unsigned short v[8];
    for (int e = 0; e < 8; ++e) v[e] = pat(iter, row, col + e);
    *(uint4*)(X + (size_t)row * CH + col) = *(uint4*)v;
  }
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(&tm[iter & 1023].w_end_max, wall_clock64());
}

// R: grid = IMGS x (CH/64) slabs; block reads 576 rows x 64 channels (128 B per row, stride 2 KB)
template <int MODE>  // 0 plain loads, 1 LDS-DMA, 2 plain loads + in-place write-back of ~value
__global__ __launch_bounds__(256) void reader(unsigned short* X, unsigned iter, Times* tm, unsigned* bad, unsigned* first_bad) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[32 * 128];
  if (threadIdx.x == 0) atomicMin(&tm[iter & 1023].r_start_min, wall_clock64());
  const int img = blockIdx.x / (CH / 64), slab = blockIdx.x % (CH / 64);
  const int c8 = threadIdx.x & 7, r0 = threadIdx.x >> 3;  // 32 rows per pass
  unsigned nbad = 0, seen = 0;
  for (int p = 0; p < PIX / 32; ++p) {
    const int row = img * PIX + p * 32 + r0, col = slab * 64 + c8 * 8;
    unsigned short* src = X + (size_t)row * CH + col;
    uint4 v;
    if (MODE == 1) {
#if defined(__HIP_DEVICE_COMPILE__)
      const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(lds + wave * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      v = *(const uint4*)(lds + threadIdx.x * 16);
      __syncthreads();
#endif
    } else {
      v = *(const uint4*)src;
    }
    const unsigned short* h = (const unsigned short*)&v;
    for (int e = 0; e < 8; ++e) {
      const unsigned short want = pat(iter, row, col + e);
      if (h[e] != want) { ++nbad; seen = h[e]; }
    }
    if (MODE == 2) {
      uint4 w = v;
      w.x = ~w.x; w.y = ~w.y; w.z = ~w.z; w.w = ~w.w;
      *(uint4*)src = w;
    }
  }
  if (nbad) {
    atomicAdd(bad, nbad);
    if (atomicCAS(first_bad, 0u, iter + 1) == 0u) { first_bad[1] = blockIdx.x; first_bad[2] = seen; first_bad[3] = threadIdx.x; }
  }
}

// stream B: HBM + LDS traffic on every CU
__global__ __launch_bounds__(256) void noise(const uint4* src, uint4* dst, size_t n, int rounds) {
  __shared__ uint4 s[1024];
  uint4 acc = {0, 0, 0, 0};
  for (int r = 0; r < rounds; ++r)
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
      uint4 v = src[i];
      s[threadIdx.x * 4 + (r & 3)] = v;
      __syncthreads();
      const uint4 u = s[((threadIdx.x + 37) & 255) * 4 + (r & 3)];
      acc.x ^= u.x + v.y; acc.y += u.z; acc.z ^= v.w; acc.w += u.w;
      __syncthreads();
    }
  dst[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
static void run(const char* name, int iters, bool with_noise, hipStream_t sa, hipStream_t sb, unsigned short* X, Times* tm, unsigned* bad,
                uint4* nsrc, uint4* ndst, size_t nn) {
  CHK(hipMemset(bad, 0, 32));
  std::vector<Times> init(1024, Times{0ull, ~0ull});
  CHK(hipMemcpy(tm, init.data(), sizeof(Times) * 1024, hipMemcpyHostToDevice));
  CHK(hipDeviceSynchronize());
  const int tiles = (ROWS / 128) * (CH / 128);
  unsigned overlaps = 0;
  std::vector<Times> got(1024);
  for (int i = 0; i < iters; ++i) {
    if (with_noise && (i % 4) == 0) hipLaunchKernelGGL(noise, dim3(2048), dim3(256), 0, sb, nsrc, ndst, nn, 2);
    const unsigned delay_tile = (unsigned)((i * 7919u) % tiles);
    hipLaunchKernelGGL(writer, dim3(tiles), dim3(256), 0, sa, X, (unsigned)i, tm, delay_tile, (i % 3) ? 600 : 0);
    hipLaunchKernelGGL((reader<MODE>), dim3(IMGS * (CH / 64)), dim3(256), 0, sa, X, (unsigned)i, tm, bad, bad + 4);
    if ((i & 1023) == 1023 || i == iters - 1) {
      CHK(hipStreamSynchronize(sa));
      CHK(hipMemcpy(got.data(), tm, sizeof(Times) * 1024, hipMemcpyDeviceToHost));
      for (auto& t : got)
        if (t.w_end_max != 0ull && t.r_start_min != ~0ull && t.w_end_max > t.r_start_min) ++overlaps;
      CHK(hipMemcpy(tm, init.data(), sizeof(Times) * 1024, hipMemcpyHostToDevice));
    }
  }
  CHK(hipDeviceSynchronize());
  unsigned h[8];
  CHK(hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost));
  printf("%-34s noise=%d iters=%d : mismatching elements %u, iterations where W_end > R_start %u", name, (int)with_noise, iters, h[0], overlaps);
  if (h[4]) printf("  [first: iter %u block %u saw 0x%04x thread %u]", h[4] - 1, h[5], h[6], h[7]);
  printf("\n");
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  hipStream_t sa, sb;
  CHK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
  CHK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  unsigned short* X;
  Times* tm;
  unsigned* bad;
  uint4 *nsrc, *ndst;
  const size_t nn = (size_t)64 << 20 >> 4;  // 64 MB of uint4
  CHK(hipMalloc(&X, (size_t)ROWS * CH * 2));
  CHK(hipMalloc(&tm, sizeof(Times) * 1024));
  CHK(hipMalloc(&bad, 64));
  CHK(hipMalloc(&nsrc, nn * 16));
  CHK(hipMalloc(&ndst, (size_t)2048 * 256 * 16));
  CHK(hipMemset(X, 0, (size_t)ROWS * CH * 2));
  CHK(hipMemset(nsrc, 1, nn * 16));
  for (int noise_on = 0; noise_on < 2; ++noise_on) {
    run<0>("plain loads", iters, noise_on, sa, sb, X, tm, bad, nsrc, ndst, nn);
    run<1>("LDS-DMA (global_load_lds) loads", iters, noise_on, sa, sb, X, tm, bad, nsrc, ndst, nn);
    run<2>("plain loads, in-place write-back", iters, noise_on, sa, sb, X, tm, bad, nsrc, ndst, nn);
  }
  printf("done\n");
  return 0;
}
