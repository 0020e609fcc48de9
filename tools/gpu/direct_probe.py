"""Does the register-direct form of gemm_pp_kernel (transposed accumulators: swapped MFMA operands) give the same BITS as the
staged form?  dptx_debug_set_gemm_flags(1) forces the staged epilogue.  Random operands, and small-integer operands whose
sums are exact in fp32 whatever the order (separates an epilogue bug from an accumulation-order difference)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omnidata_amd.engine import load_library

lib = load_library()
st = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16
M = 32 * 577
for name, N, K, act in (("qkv-like", 2304, 768, 0), ("fc1-like + GELU", 3072, 768, 2)):
    for kind in ("random", "small integers (exact sums)"):
        g = torch.Generator(device="cuda").manual_seed(1)
        if kind == "random":
            A = torch.randn(M, K, device="cuda", generator=g).to(bf)
            W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(bf)
        else:
            A = torch.randint(-2, 3, (M, K), device="cuda", generator=g).to(bf)
            W = torch.randint(-2, 3, (N, K), device="cuda", generator=g).to(bf)
        bias = torch.randn(N, device="cuda", generator=g)
        out = []
        for flags in (0, 1, 3):
            lib.dptx_debug_set_gemm_flags(flags)
            C = torch.zeros(M, N, device="cuda", dtype=bf)
            lib.dptx_op_gemm(0, A.data_ptr(), W.data_ptr(), bias.data_ptr(), None, C.data_ptr(), M, N, K, act, 0, 0, 0, st)
            torch.cuda.synchronize()
            out.append(C)
        lib.dptx_debug_set_gemm_flags(0)
        ref = (A.float() @ W.float().t() + bias)
        if act == 2:
            ref = torch.nn.functional.gelu(ref)
        d01 = (out[0].view(torch.int16) != out[1].view(torch.int16)).sum().item()
        d13 = (out[1].view(torch.int16) != out[2].view(torch.int16)).sum().item()
        e0 = (out[0].float() - ref).abs().max().item()
        e1 = (out[1].float() - ref).abs().max().item()
        print(f"{name}, {kind}: direct vs staged {d01} of {M * N} outputs differ; staged persistent vs per-tile {d13};"
              f" max|err| vs fp32 torch: direct {e0:.3e}, staged {e1:.3e}")
