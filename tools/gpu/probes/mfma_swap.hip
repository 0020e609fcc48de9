// Build: hipcc --offload-arch=gfx950 -O2 -o tools/gpu/probes/mfma_swap tools/gpu/probes/mfma_swap.hip (the executable is git-ignored).
// Is v_mfma_f32_32x32x16_bf16 bit-symmetric under swapping its operands?  D = A B with A 32x16, B 16x32; the swapped call
// computes D' = B^T A^T = D^T from the SAME registers (a lane's fragment is row (lane % 32), k = 8 (lane / 32) .. +7 of its
// matrix either way).  Compares D'[n][m] with D[m][n] bit for bit over random operands with a wide exponent spread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const uint16_t* a, const uint16_t* b, float* d1, float* d2, int chain) {
  const int lane = threadIdx.x, lr = lane & 31, lh = lane >> 5;
  f32x16 z1, z2;
  for (int r = 0; r < 16; ++r) { z1[r] = 0.f; z2[r] = 0.f; }
  for (int c = 0; c < chain; ++c) {   // `chain` k-steps accumulated, like a k-loop
    bf16x8 fa, fb;
    for (int e = 0; e < 8; ++e) {
      fa[e] = (short)a[(c * 32 + lr) * 16 + lh * 8 + e];   // A[m = lr][k]
      fb[e] = (short)b[(c * 32 + lr) * 16 + lh * 8 + e];   // B^T[n = lr][k]
    }
    z1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, z1, 0, 0, 0);   // D[m][n]: lane = n, regs = m
    z2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, z2, 0, 0, 0);   // D'[n][m]: lane = m, regs = n
  }
  for (int r = 0; r < 16; ++r) {
    const int idx = (r & 3) + 8 * (r >> 2) + 4 * lh;
    d1[idx * 32 + lr] = z1[r];       // D[m = idx][n = lr]
    d2[lr * 32 + idx] = z2[r];       // D'[n = idx][m = lr] stored as D[m][n]
  }
}
static uint16_t bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
int main() {
  const int chain = 12;
  uint16_t *a, *b; float *d1, *d2;
  hipMallocManaged(&a, chain * 512 * 2); hipMallocManaged(&b, chain * 512 * 2);
  hipMallocManaged(&d1, 4096); hipMallocManaged(&d2, 4096);
  int bad_total = 0;
  for (int trial = 0; trial < 200; ++trial) {
    srand(trial);
    for (int i = 0; i < chain * 512; ++i) {
      const float sa = ldexpf((rand() / (float)RAND_MAX) * 2.f - 1.f, (rand() % 17) - 8);
      const float sb = ldexpf((rand() / (float)RAND_MAX) * 2.f - 1.f, (rand() % 17) - 8);
      a[i] = bf16(sa); b[i] = bf16(sb);
    }
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d1, d2, chain);
    hipDeviceSynchronize();
    int bad = 0;
    for (int i = 0; i < 1024; ++i) bad += memcmp(&d1[i], &d2[i], 4) != 0;
    if (bad && bad_total < 3) {
      for (int i = 0; i < 1024; ++i) if (memcmp(&d1[i], &d2[i], 4)) { printf("trial %d: D[%d][%d] = %.9g vs swapped %.9g\n", trial, i / 32, i % 32, d1[i], d2[i]); break; }
    }
    bad_total += bad;
  }
  printf("mfma_f32_32x32x16_bf16, %d chained k-steps, 200 trials: %d of %d outputs differ under operand swap\n", chain, bad_total, 200 * 1024);
  return 0;
}
