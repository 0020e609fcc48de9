// Probe (round 6): semantics of v_cvt_scalef32_pk_bf8_f16 and of the e5m2 format / E8M0 scale of
// v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950.   hipcc --offload-arch=gfx950 -O2 bf8_probe.hip -o bf8_probe && ./bf8_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef short s16x2_t __attribute__((ext_vector_type(2)));
typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

__global__ void cvt_k(const _Float16* in, unsigned char* out, float scale, int n) {
  const int i = threadIdx.x;
  if (2 * i + 1 >= n + 1) return;
  f16x2_t v = {in[2 * i], in[2 * i + 1]};
  s16x2_t r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_bf8_f16(r, v, scale, false);
  out[2 * i] = (unsigned char)(r.x & 0xff);
  out[2 * i + 1] = (unsigned char)((r.x >> 8) & 0xff);
}

__global__ void mfma_k(float* out, int a_byte, int b_byte, int sa, int sb) {
  const int v = a_byte | (a_byte << 8) | (a_byte << 16) | (a_byte << 24);
  const int w = b_byte | (b_byte << 8) | (b_byte << 16) | (b_byte << 24);
  i32x8_t a = {v, v, v, v, v, v, v, v}, b = {w, w, w, w, w, w, w, w};
  f32x16_t c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 1, 0, sa, 0, sb);
  if (threadIdx.x == 0) out[0] = c[0];
  if (threadIdx.x == 37) out[1] = c[5];
}

static float bf8_to_float(unsigned char b) {
  unsigned short h = (unsigned short)b << 8;
  _Float16 f;
  memcpy(&f, &h, 2);
  return (float)f;
}

int main() {
  const float vals[] = {1.0f, 1.125f, 1.375f, 1.625f, 1.875f, -3.3f, 0.1f, 1e-3f, 6e-5f, 3e-5f, 1e-5f, 3e-6f, 6e-8f, 57344.f, 60000.f, 61440.f, 65504.f, 0.f, 448.f, 500.f};
  const int n = sizeof(vals) / sizeof(vals[0]);
  _Float16 h[32];
  for (int i = 0; i < n; ++i) h[i] = (_Float16)vals[i];
  _Float16* din; unsigned char* dout; float* df;
  hipMalloc(&din, 64); hipMalloc(&dout, 64); hipMalloc(&df, 16);
  hipMemcpy(din, h, n * 2, hipMemcpyHostToDevice);
  const float scales[] = {1.0f, 2.0f, 0.5f, 1.0f / 1024.0f, 1024.0f, 3.0f};
  for (float sc : scales) {
    hipLaunchKernelGGL(cvt_k, dim3(1), dim3(64), 0, 0, din, dout, sc, n);
    unsigned char o[64];
    hipMemcpy(o, dout, 64, hipMemcpyDeviceToHost);
    printf("scale %g:\n", sc);
    for (int i = 0; i < n; ++i) printf("  %12g -> 0x%02x = %g\n", (float)h[i], o[i], bf8_to_float(o[i]));
  }
  const int cases[][4] = {{0x3c, 0x3c, 0x7f7f7f7f, 0x7f7f7f7f}, {0x3c, 0x3c, 0x76767676, 0x7f7f7f7f}, {0x3c, 0x40, 0x7f7f7f7f, 0x7e7e7e7e},
                          {0x01, 0x3c, 0x7f7f7f7f, 0x7f7f7f7f}, {0x01, 0x01, 0x7f7f7f7f, 0x7f7f7f7f}, {0x7b, 0x3c, 0x7f7f7f7f, 0x7f7f7f7f}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(mfma_k, dim3(1), dim3(64), 0, 0, df, c[0], c[1], c[2], c[3]);
    float o[2];
    hipMemcpy(o, df, 8, hipMemcpyDeviceToHost);
    printf("mfma e5m2 a=0x%02x (%g) b=0x%02x (%g) scale_a=0x%02x scale_b=0x%02x -> %g %g   (64 a b = %g)\n", c[0], bf8_to_float(c[0]), c[1],
           bf8_to_float(c[1]), c[2] & 0xff, c[3] & 0xff, o[0], o[1], 64.0 * bf8_to_float(c[0]) * bf8_to_float(c[1]));
  }
  return 0;
}
