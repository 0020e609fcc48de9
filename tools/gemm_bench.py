"""Per-shape micro-benchmark of the implicit-GEMM kernel on the DPT-Hybrid layer shapes (B=32).
Usage (GPU box): python tools/gemm_bench.py [--dtype bf16] ; DPTX_GEMM=reg selects the register-staged variant."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnidata_amd.build import build  # noqa: E402
from omnidata_amd.engine import DTYPES, load_library  # noqa: E402

B = 32
DENSE = [("vit.qkv", B * 577, 2304, 768), ("vit.proj", B * 577, 768, 768), ("vit.fc1", B * 577, 3072, 768),
         ("vit.fc2", B * 577, 768, 3072), ("stem.gemm", B * 36864, 64, 192), ("patch.proj", B * 576, 768, 1024),
         # calibration against the square-GEMM figures of the CDNA4 programming guide (not DPT shapes; --only cal.4096,...)
         ("cal.4096", 4096, 4096, 4096), ("cal.8192", 8192, 8192, 8192)]
# name, H, Cin, Cout, k, stride, pad, Ho
CONV = [("rcu@96", 96, 256, 256, 3, 1, 1, 96), ("rcu@48", 48, 256, 256, 3, 1, 1, 48), ("rcu@24", 24, 256, 256, 3, 1, 1, 24),
        ("rcu@12", 12, 256, 256, 3, 1, 1, 12), ("head.0", 192, 256, 128, 3, 1, 1, 192), ("head.2", 384, 128, 32, 3, 1, 1, 384),
        ("l2_rn", 48, 512, 256, 3, 1, 1, 48), ("l3_rn", 24, 768, 256, 3, 1, 1, 24), ("out_conv@96", 96, 256, 256, 1, 1, 0, 96),
        ("s0.c1", 96, 256, 64, 1, 1, 0, 96), ("s0.c2", 96, 64, 64, 3, 1, 1, 96), ("s0.c3", 96, 64, 256, 1, 1, 0, 96),
        ("s1.c1", 48, 512, 128, 1, 1, 0, 48), ("s1.c2", 48, 128, 128, 3, 1, 1, 48), ("s1.c3", 48, 128, 512, 1, 1, 0, 48),
        ("s2.c1", 24, 1024, 256, 1, 1, 0, 24), ("s2.c2", 24, 256, 256, 3, 1, 1, 24), ("s2.c3", 24, 256, 1024, 1, 1, 0, 24),
        ("s1.b0.c2(s2)", 96, 128, 128, 3, 2, 0, 48), ("pp4.conv2", 24, 768, 768, 3, 2, 1, 12)]


ITERS = [10]


def timeit(fn, iters=None):
    iters = iters or ITERS[0]
    return _timeit(fn, iters)


def _timeit(fn, iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--only", default=None, help="comma separated shape names")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="images (conv shapes only): 16 = what one of the two streams launches")
    args = ap.parse_args()
    global B
    B = args.batch
    only = set(args.only.split(",")) if args.only else None
    ITERS[0] = args.iters
    build()
    lib = load_library()
    fp8 = args.dtype == "fp8"
    dt, tdt = DTYPES[args.dtype], (torch.bfloat16 if args.dtype in ("bf16", "fp8") else torch.float16)
    st = torch.cuda.current_stream().cuda_stream
    print(f"variant={os.environ.get('DPTX_GEMM', 'glds')} dtype={args.dtype}")
    tot_ms, tot_flop = 0.0, 0.0
    # the ViT GEMMs run with the epilogue the engine gives them: fc1 bias + GELU; proj / fc2 bias + in-place fp32 residual
    EPI = {"vit.fc1": (2, False), "vit.proj": (0, True), "vit.fc2": (0, True)}
    for name, M, N, K in DENSE:
        if (only and name not in only) or fp8 or (name.startswith("cal.") and not only):
            continue
        act, inplace32 = EPI.get(name, (0, False))
        A = torch.randn(M, K, device="cuda").to(tdt)
        W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(tdt)
        C = torch.zeros(M, N, device="cuda", dtype=torch.float32 if inplace32 else tdt)
        bias = torch.randn(N, device="cuda") * (0.0 if inplace32 else 1.0)
        R = C.data_ptr() if inplace32 else None
        ms = timeit(lambda: lib.dptx_op_gemm(dt, A.data_ptr(), W.data_ptr(), bias.data_ptr(), R, C.data_ptr(), M, N, K, act, 0,
                                             int(inplace32), int(inplace32), st))
        fl = 2.0 * M * N * K
        print(f"{name:14s} M={M:8d} N={N:5d} K={K:5d}  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TF/s")
        tot_ms += ms; tot_flop += fl
    if fp8:  # the ViT linears on the e4m3 kernel (DPTX_FLAG_FP8_VIT): a dense GEMM is a 1x1 convolution over a 577 x 1 "image"
        for name, M, N, K in DENSE:
            if (only and name not in only) or not name.startswith("vit.") or name == "vit.proj":
                continue
            X = torch.randn(B, 577, 1, K, device="cuda").to(torch.float8_e4m3fn)
            Wt = (torch.randn(N, 1, 1, K, device="cuda") * 16.0).to(torch.float8_e4m3fn)
            Y = torch.empty(B, 577, 1, N, device="cuda", dtype=tdt)
            Y8 = torch.empty(B, 577, 1, N, device="cuda", dtype=torch.uint8)
            bias = torch.randn(N, device="cuda")
            act = 2 if name == "vit.fc1" else 0
            ms = timeit(lambda: lib.dptx_op_conv_fp8(X.data_ptr(), Wt.data_ptr(), bias.data_ptr(), None, Y.data_ptr(), Y8.data_ptr(), B, 577, 1,
                                                     K, N, 1, 1, 0, 0, 577, 1, act, 0, 1.0 / 768, st))
            fl = 2.0 * B * 577 * N * K
            print(f"{name + '@fp8':14s} M={B * 577:8d} N={N:5d} K={K:5d}  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TF/s")
            tot_ms += ms; tot_flop += fl
    for name, H, Cin, Cout, k, s, pad, Ho in CONV:
        if only and name not in only:
            continue
        if fp8 and Cin % 128:
            continue
        X = torch.randn(B, H, H, Cin, device="cuda").to(torch.float8_e4m3fn if fp8 else tdt)
        Wt = (torch.randn(Cout, k, k, Cin, device="cuda") * (16.0 if fp8 else (k * k * Cin) ** -0.5)).to(torch.float8_e4m3fn if fp8 else tdt)
        Y = torch.empty(B, Ho, Ho, Cout, device="cuda", dtype=tdt)
        bias = torch.randn(Cout, device="cuda")
        if fp8:  # the fp8 dtype's convolution: e4m3 operands, bf16 output + e4m3 copy of it
            Y8 = torch.empty(B, Ho, Ho, Cout, device="cuda", dtype=torch.uint8)
            ms = timeit(lambda: lib.dptx_op_conv_fp8(X.data_ptr(), Wt.data_ptr(), bias.data_ptr(), None, Y.data_ptr(), Y8.data_ptr(), B, H,
                                                     H, Cin, Cout, k, s, pad, pad, Ho, Ho, 0, 0, 1.0 / 768, st))
        else:
            ms = timeit(lambda: lib.dptx_op_conv(dt, X.data_ptr(), Wt.data_ptr(), bias.data_ptr(), None, Y.data_ptr(), B, H, H, Cin, Cout,
                                                 k, s, pad, pad, Ho, Ho, 0, 0, st))
        M, K = B * Ho * Ho, k * k * Cin
        fl = 2.0 * M * Cout * K
        print(f"{name:14s} M={M:8d} N={Cout:5d} K={K:5d}  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TF/s")
        tot_ms += ms; tot_flop += fl
        del X, Wt, Y
    print(f"TOTAL {tot_ms:.3f} ms  {tot_flop / tot_ms / 1e9:.1f} TF/s (unweighted list)")


if __name__ == "__main__":
    main()
