// common.h -- shared device helpers for the dptx kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dptx {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));  // native 16-B register value (HIP's uint4
                                                                // is a struct: arrays of it can end up in scratch)
typedef short s16x2_t __attribute__((ext_vector_type(2)));

constexpr int DT_BF16 = 0;
constexpr int DT_FP16 = 1;
constexpr int DT_FP8 = 2;  // OCP e4m3 MFMA operands (GEMM A / W only; everything stored stays bf16 + an fp8 copy)

// Precision modes of the engine ("dtype" in the C ABI):
//   0 bf16, 1 fp16 : one 16-bit value per element, one MFMA per product.
//   2 bf16x3       : every 16-bit tensor is stored as TWO bf16 planes hi = bf16(x), lo = bf16(x - hi)
//                    (16 significand bits); a product is A_hi*W_hi + A_lo*W_hi + A_hi*W_lo = 3 MFMAs
//                    with fp32 accumulation.  The lo plane of a tensor lives at a fixed distance from
//                    its hi plane: `Planes::act` elements for activations (second half of the arena),
//                    `Planes::w` for packed GEMM weights (second half of the blob).
//   3 fp16x3       : the same with fp16 planes: hi = fp16(x) is at the same time the operand of a single-pass fp16
//                    layer, so one engine can run some layer groups with 3 MFMAs per product and the others with 1
//                    ("mixed" dtype of the C ABI: dptx_config.x3_groups) without converting tensors in between.
//                    gfx950's f16 MFMA keeps subnormal inputs (tests/test_gpu_mixed.py pins that), so lo planes that
//                    fall below 2^-14 still carry their bits (spacing 2^-24).
//   4 fp8 (GEMM only): A and W are OCP e4m3 bytes (128 of them per 128-byte k-tile row: half the bytes per flop of the
//                    16-bit kernels), products on the block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 with unit scales
//                    (2x the bf16 MFMA rate), fp32 accumulate; C / residuals bf16.  The engine's "fp8" dtype runs the
//                    decoder's RCU / out_conv / head convolutions this way and everything else in bf16.
constexpr int MODE_BF16 = 0, MODE_FP16 = 1, MODE_BF16X3 = 2, MODE_FP16X3 = 3, MODE_FP8 = 4;
__host__ __device__ constexpr bool mode_is_x3(int mode) { return mode == MODE_BF16X3 || mode == MODE_FP16X3; }
#define DPTX_DISPATCH_MODE(mode, ...)                                         \
  switch (mode) {                                                             \
    case ::dptx::MODE_BF16:   { constexpr int DT = ::dptx::DT_BF16, PL = 1; __VA_ARGS__; } break; \
    case ::dptx::MODE_FP16:   { constexpr int DT = ::dptx::DT_FP16, PL = 1; __VA_ARGS__; } break; \
    case ::dptx::MODE_BF16X3: { constexpr int DT = ::dptx::DT_BF16, PL = 2; __VA_ARGS__; } break; \
    case ::dptx::MODE_FP16X3: { constexpr int DT = ::dptx::DT_FP16, PL = 2; __VA_ARGS__; } break; \
    default: return hipErrorInvalidValue;                                     \
  }
struct Planes {
  long long act;  // element (uint16) distance hi -> lo plane of an activation tensor
  long long w;    // same for packed 16-bit weights
};

// 16-bit storage/MFMA-operand type traits.  DT = 0: bf16, 1: fp16.
template <int DT> struct T16;

template <> struct T16<DT_BF16> {
  static __device__ __forceinline__ float tof(uint16_t u) { return __uint_as_float(((uint32_t)u) << 16); }
  static __device__ __forceinline__ uint16_t fromf(float f) {
    __bf16 h = (__bf16)f;  // RNE (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(uint16_t, h);
  }
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    bf16x2_t v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
  }
  static __device__ __forceinline__ f32x16_t mfma32(const uint4& a, const uint4& b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16_t mfma32(u32x4_t a, u32x4_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                   __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};

template <> struct T16<DT_FP16> {
  static __device__ __forceinline__ float tof(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
  // The converted bits go through an empty asm: hipcc (fp-contract=fast) otherwise folds `x - tof(fromf(x))` with a
  // producing multiply into v_fma_mix*_f16, which rounds the EXACT product to fp16, while the stored hi value is the
  // fp16 rounding of the fp32-rounded product (v_cvt_pk_f16_f32) -- the two differ on fp16 ties (2^-14 of all values),
  // and the hi/lo pair of an fp16x3 tensor then sums to a value one fp16 ulp off.  (Found on the attention output:
  // 58 of 886 272 elements, each off by exactly 2^-13 or 2^-14.)
  static __device__ __forceinline__ uint16_t fromf(float f) {
    _Float16 h = (_Float16)f;
    uint32_t r = __builtin_bit_cast(uint16_t, h);
    asm("" : "+v"(r));
    return (uint16_t)r;
  }
  static __device__ __forceinline__ uint32_t pack2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
    f16x2_t v = {(_Float16)a, (_Float16)b};
    uint32_t r = __builtin_bit_cast(uint32_t, v);
    asm("" : "+v"(r));
    return r;
  }
  static __device__ __forceinline__ f32x16_t mfma32(const uint4& a, const uint4& b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a),
                                                  __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16_t mfma32(u32x4_t a, u32x4_t b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a),
                                                  __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

// unpack 8 x 16-bit (one uint4) to 8 floats
template <int DT>
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = T16<DT>::tof((uint16_t)(w[i] & 0xffffu));
    f[2 * i + 1] = T16<DT>::tof((uint16_t)(w[i] >> 16));
  }
}
template <int DT>
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = T16<DT>::pack2(f[0], f[1]);
  v.y = T16<DT>::pack2(f[2], f[3]);
  v.z = T16<DT>::pack2(f[4], f[5]);
  v.w = T16<DT>::pack2(f[6], f[7]);
  return v;
}

// ReLU on two packed 16-bit floats (bf16 or fp16).  Both are sign-magnitude, so as int16 a
// negative float is a negative integer and a positive float keeps its bits: max(x, 0) as packed
// int16 is exactly ReLU (one v_pk_max_i16).
__device__ __forceinline__ uint32_t relu2(uint32_t v) {
  const s16x2_t z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, v), z));
}
__device__ __forceinline__ u32x4_t relu8(u32x4_t v) {
  v.x = relu2(v.x); v.y = relu2(v.y); v.z = relu2(v.z); v.w = relu2(v.w);
  return v;
}
__device__ __forceinline__ uint4 relu8(uint4 v) {
  v.x = relu2(v.x); v.y = relu2(v.y); v.z = relu2(v.z); v.w = relu2(v.w);
  return v;
}

// erf-GELU (timm Mlp / ProjectReadout use nn.GELU(), the exact erf form): gelu_erf() below.
// 8 consecutive elements <-> floats, with optional hi/lo planes
template <int DT, int PL>
__device__ __forceinline__ void load8f(const uint16_t* p, long long plane, float* f) {
  unpack8<DT>(*(const uint4*)p, f);
  if (PL == 2) {
    float g[8];
    unpack8<DT>(*(const uint4*)(p + plane), g);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += g[e];
  }
}
template <int DT, int PL>
__device__ __forceinline__ void store8f(uint16_t* p, long long plane, const float* f) {
  const uint4 hi = pack8<DT>(f);
  *(uint4*)p = hi;
  if (PL == 2) {
    float h[8], l[8];
    unpack8<DT>(hi, h);
#pragma unroll
    for (int e = 0; e < 8; ++e) l[e] = f[e] - h[e];
    *(uint4*)(p + plane) = pack8<DT>(l);
  }
}
// 8 x 16-bit in registers (hi plane, lo plane when PL == 2) -> 8 floats
template <int DT, int PL>
__device__ __forceinline__ void unpack8x(u32x4_t hi, u32x4_t lo, float* f) {
  uint4 h;
  h.x = hi.x; h.y = hi.y; h.z = hi.z; h.w = hi.w;
  unpack8<DT>(h, f);
  if (PL == 2) {
    uint4 l;
    l.x = lo.x; l.y = lo.y; l.z = lo.z; l.w = lo.w;
    float g[8];
    unpack8<DT>(l, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] += g[e];
  }
}
// ReLU of a hi/lo pair: sign(x) == sign(hi), so lo is zeroed wherever hi is negative.
// Whole-vector form (an element-indexed loop over a vector reference was miscompiled by hipcc
// 7.2 into "element 0 broadcast to all four dwords").
typedef short s16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void relu8_planes(u32x4_t& hi, u32x4_t& lo) {
  const s16x8_t h = __builtin_bit_cast(s16x8_t, hi);
  const s16x8_t neg = h >> 15;  // 0xFFFF where negative
  const s16x8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
  lo = __builtin_bit_cast(u32x4_t, (s16x8_t)(__builtin_bit_cast(s16x8_t, lo) & ~neg));
  hi = __builtin_bit_cast(u32x4_t, __builtin_elementwise_max(h, zero));
}

// 8 floats -> 8 OCP e4m3 bytes (RNE, saturating at +-448); v_cvt_pk_fp8_f32 packs two floats into one half of a dword
__device__ __forceinline__ uint2 pack_fp8x8(const float* f) {
  float c[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) c[e] = fminf(fmaxf(f[e], -448.f), 448.f);
  int lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
  return make_uint2((uint32_t)lo, (uint32_t)hi);
}

// caller-side image / result buffers (include/dptx.h DPTX_IO_*): 0 fp32, 1 bf16, 2 fp16
constexpr int IO_FP32 = 0, IO_BF16 = 1, IO_FP16 = 2;
__device__ __forceinline__ float io_load(const void* p, long long i, int io) {
  if (io == IO_FP32) return ((const float*)p)[i];
  const uint16_t u = ((const uint16_t*)p)[i];
  return io == IO_BF16 ? T16<DT_BF16>::tof(u) : T16<DT_FP16>::tof(u);
}
__device__ __forceinline__ void io_store(void* p, long long i, float v, int io) {
  if (io == IO_FP32) ((float*)p)[i] = v;
  else ((uint16_t*)p)[i] = io == IO_BF16 ? T16<DT_BF16>::fromf(v) : T16<DT_FP16>::fromf(v);
}

// bilinear blend of a 2x2 neighbourhood in ATen's association, ly0*(lx0*a + lx1*b) + ly1*(lx0*c + lx1*d), with the
// fused-multiply-adds spelled out so that every kernel that interpolates rounds identically
__device__ __forceinline__ float bilerp(float a, float b, float c, float d, float lx0, float lx1, float ly0, float ly1) {
  const float t0 = __fmaf_rn(lx1, b, __fmul_rn(lx0, a));
  const float t1 = __fmaf_rn(lx1, d, __fmul_rn(lx0, c));
  return __fmaf_rn(ly1, t1, __fmul_rn(ly0, t0));
}

// erfc via Abramowitz & Stegun 7.1.28, erfc(z) = (1 + a1 z + ... + a6 z^6)^-16 for z >= 0 (|abs err| <= 3e-7; measured
// 7.1e-7 on gelu over [-12, 12] in fp32, the same league as 7.1.26's 4.7e-7, which round 1-2 used): six fused
// multiply-adds, four squarings and ONE quarter-rate instruction (v_rcp_f32) -- 7.1.26 needs v_rcp_f32 + v_exp_f32 and as
// many multiply-adds; everything but the reciprocal packs into v_pk_*_f32.  The fc1 epilogue evaluates this for 56.7 M
// elements per launch with no MFMA running beside it (~10 us of a 24 us tile with 7.1.26).
// gelu(x) = x/2 * (1 + erf(x/sqrt 2)) = (h + |h|) - |h| * erfc(|x|/sqrt 2), h = x/2: exact 0 - |h| erfc for x < 0 (no
// cancellation in the tail); p^16 overflows to inf beyond |x| ~ 24, whose reciprocal is the right limit 0.
// Two elements at a time on 2-vectors: hipcc turns these into v_pk_mul_f32 / v_pk_fma_f32 (a scalar form becomes
// v_fmaak_f32 with literal constants, one element per instruction).
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_erf2(float& x0, float& x1) {
  // every multiply-add is an explicit fma: the value must not depend on what -ffp-contract does at a call site
  auto sp = [](float c) { return f32x2_t{c, c}; };
  const f32x2_t x = {x0, x1};
  const f32x2_t h = x * 0.5f;
  const f32x2_t ha = {fabsf(h.x), fabsf(h.y)};
  const f32x2_t z = ha * 1.41421356237309504880f;  // |x| / sqrt 2
  f32x2_t q = __builtin_elementwise_fma(z, sp(0.0000430638f), sp(0.0002765672f));
  q = __builtin_elementwise_fma(q, z, sp(0.0001520143f));
  q = __builtin_elementwise_fma(q, z, sp(0.0092705272f));
  q = __builtin_elementwise_fma(q, z, sp(0.0422820123f));
  q = __builtin_elementwise_fma(q, z, sp(0.0705230784f));
  q = __builtin_elementwise_fma(q, z, sp(1.0f));
  q *= q; q *= q; q *= q; q *= q;
  const f32x2_t r = {__builtin_amdgcn_rcpf(q.x), __builtin_amdgcn_rcpf(q.y)};
  const f32x2_t y = __builtin_elementwise_fma(-ha, r, h + ha);
  x0 = y.x;
  x1 = y.y;
}

// Bijective XCD-aware remap of a 1-D block id: blocks that the dispatcher places on one
// XCD (id % 8) receive a contiguous range of logical work-group ids, so neighbouring tiles
// share that XCD's L2 (cdna_hip_programming.md T1, bijective form).
__device__ __forceinline__ int xcd_remap(int id, int nwg) {
  const int xcd = id & 7, idx = id >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace dptx
