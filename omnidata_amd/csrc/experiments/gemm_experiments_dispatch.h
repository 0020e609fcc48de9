#pragma once
// experiments/gemm_experiments_dispatch.h -- host-side selection of the experiment kernels (gemm_experiments.h).  Only in
// builds made with -DDPTX_EXPERIMENTS (build.py: DPTX_CXXFLAGS=-DDPTX_EXPERIMENTS DPTX_LIB_SUFFIX=_exp); the default
// libdptx.so neither contains these kernels nor reads these environment variables:
//   DPTX_PP=0    lockstep 256x256 loop (gemm_glds_kernel) instead of the ping-pong kernel
//   DPTX_PP=2    phased ("8-phase") 256x256 kernel (bf16 only)
//   DPTX_PP=7    256x128 three-stage kernel wherever the 256x256 rule would fire (bf16 only)
//   DPTX_HALO=1  halo-resident 3x3 convolution on the large decoder maps
static int experiment_pp() {
  static int pp = -1;
  if (pp < 0) { const char* t = getenv("DPTX_PP"); pp = t ? atoi(t) : 1; }
  return pp;
}

template <int DT>
static hipError_t launch_ph(const GemmParams& q, int tiles, size_t smem, hipStream_t stream) {
  auto go = [&](auto k) {
    set_smem_attr(k, smem);
    hipLaunchKernelGGL(k, dim3(tiles), dim3(512), smem, stream, q);
  };
  if (q.a_relu) go(gemm_ph_kernel<DT, true>);
  else go(gemm_ph_kernel<DT, false>);
  return hipGetLastError();
}

// Shapes gemm_halo_kernel accepts: 3x3 / stride 1 / pad 1 on a dense NHWC map whose width is a multiple of 32 and height a
// multiple of 8, Cin % 64 == 0, N % 256 == 0, 16-bit operands, no GroupNorm statistics.  launch_gemm forces k_tap_fast for
// them whichever kernel ends up running, so that the k order -- and with it every bit of the result -- is the same.
static bool halo_shape(const GemmParams& p) {
  static int on = -1;
  if (on < 0) { const char* t = getenv("DPTX_HALO"); on = (t && t[0] == '1') ? 1 : 0; }
  return on && p.ksz == 3 && p.stride == 1 && p.pad_t == 1 && p.pad_l == 1 && p.Wout == p.Win && p.a_rpi == p.Hin * p.Win &&
         p.Win % 32 == 0 && p.Hin % 8 == 0 && p.Cin % 64 == 0 && p.N % 256 == 0 && p.K == 9 * p.Cin && !p.a_fp32 &&
         p.gn_part == nullptr && p.M % p.a_rpi == 0 && p.c_rpi == 0x7fffffff && p.a_bytes > 0 && p.a_bytes < (1ll << 31) &&
         (long long)p.N * p.ldw * 2 < (1ll << 31);
}

template <int DT>
static hipError_t launch_halo(const GemmParams& p, hipStream_t stream) {
  const int tiles_m = (p.M / p.a_rpi) * (p.Win / 32) * (p.Hin / 8), tiles_n = p.N / 256;
  GemmParams q = p;
  choose_xcd_grid(p, tiles_m, tiles_n, q.xcd_m, q.xcd_n);
  const int tiles = 8 * ((tiles_m + q.xcd_m - 1) / q.xcd_m) * (tiles_n / q.xcd_n);
  constexpr size_t smem = 2 * 32 * 1024 + 2 * 44 * 1024;
  auto go = [&](auto k) {
    set_smem_attr(k, smem);
    hipLaunchKernelGGL(k, dim3(tiles), dim3(512), smem, stream, q);
  };
  if (p.a_relu) go(gemm_halo_kernel<DT, true>);
  else go(gemm_halo_kernel<DT, false>);
  return hipGetLastError();
}

// returns true when an experiment kernel took the launch (result in r)
template <int DT, int PL>
static bool launch_experiment(const GemmParams& p, int forced, long long m256, hipStream_t stream, hipError_t& r) {
  if constexpr (PL == 1 && DT != DT_FP8) {
    if (halo_shape(p) && forced == 0 && (long long)(p.M / p.a_rpi) * (p.Win / 32) * (p.Hin / 8) * (p.N / 256) >= 200) {
      r = launch_halo<DT>(p, stream);
      return true;
    }
  }
  if constexpr (PL == 1 && DT == DT_BF16) {
    const bool ok7 = !p.a_fp32 && p.a_bytes > 0 && p.a_bytes < (1ll << 31) && p.M < (1 << 23) &&
                     (long long)p.N * p.ldw * 2 < (1ll << 31) && p.N % 128 == 0 && p.K >= 512 && p.gn_part == nullptr;
    if (experiment_pp() == 7 && ok7 && m256 * (p.N / 128) >= 200) {
      const int tiles_m = (int)m256, tiles_n = p.N / 128;
      GemmParams q = p;
      choose_xcd_grid(p, tiles_m, tiles_n, q.xcd_m, q.xcd_n);
      const int tiles = 8 * ((tiles_m + q.xcd_m - 1) / q.xcd_m) * (tiles_n / q.xcd_n);
      constexpr size_t smem7 = 3 * 48 * 1024;
      auto go = [&](auto k) {
        set_smem_attr(k, smem7);
        hipLaunchKernelGGL(k, dim3(tiles), dim3(512), smem7, stream, q);
      };
      if (p.a_relu) go(gemm_p3_kernel<DT, true>);
      else go(gemm_p3_kernel<DT, false>);
      r = hipGetLastError();
      return true;
    }
  }
  return false;
}
