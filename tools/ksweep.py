"""T(K) = a + b*K for the implicit-GEMM kernel at fixed M, N: separates the per-tile fixed cost from the main loop."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnidata_amd.engine import load_library
from tools.gemm_bench import timeit
lib = load_library()
st = torch.cuda.current_stream().cuda_stream
for (M, N) in ((18464, 3072), (18464, 768), (294912, 256)):
    for c_fp32 in (0, 1):
        xs, ys = [], []
        for K in (64, 128, 256, 512, 768, 1536, 3072):
            A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
            C = torch.empty(M, N, device="cuda", dtype=torch.float32 if c_fp32 else torch.bfloat16)
            ms = timeit(lambda: lib.dptx_op_gemm(0, A.data_ptr(), W.data_ptr(), None, None, C.data_ptr(), M, N, K, 0, 0, c_fp32, 0, st), 20)
            xs.append(K); ys.append(ms)
        b = (ys[-1] - ys[-2]) / (xs[-1] - xs[-2])
        a = ys[-1] - b * xs[-1]
        print(f"M={M} N={N} c_fp32={c_fp32}: " + " ".join(f"K{k}:{y*1e3:.1f}us" for k, y in zip(xs, ys)) +
              f" | slope {b*1e3*64:.2f} us per k-tile row, intercept {a*1e3:.1f} us; asymptotic {2.0*M*N*64/(b*64)/1e9:.0f} TF/s")
