"""Where a tile's time goes in the persistent ping-pong kernel (gemm_pp_kernel): cycle stamps of thread 0 of block 0
(build with DPTX_CXXFLAGS=-DDPTX_TRACE DPTX_LIB_SUFFIX=_trace, run with DPTX_LIB=omnidata_amd/libdptx_trace.so).
Per tile:  T0 loop top (past the barrier) | T1 k-loop done | T2 next tile's first k-tile issued | T3 epilogue done
Epilogue:  E0 entry | E1 bias / row table done | E2 slab 0 out | E3 slabs 1-3 out"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omnidata_amd.engine import load_library

lib = load_library()
st = torch.cuda.current_stream().cuda_stream
buf = torch.zeros(8, 64, 4, dtype=torch.int64, device="cuda")


def report(name, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf.zero_()
    lib.dptx_debug_set_trace(buf.data_ptr())
    fn()
    torch.cuda.synchronize()
    lib.dptx_debug_set_trace(None)
    t = buf.cpu()[0]
    cal = t[62]
    cyc, wall = int(cal[2] - cal[0]), int(cal[3] - cal[1])
    ghz = cyc / (wall * 10.0) if wall > 0 else 0.0
    print(f"== {name}   [{ghz:.2f} cycles/ns]")
    prev_t3 = None
    for ti in range(8):
        T, E = t[40 + ti], t[48 + ti]
        if int(T[0]) == 0:
            break
        gap = int(T[0] - prev_t3) if prev_t3 is not None else 0
        print(f"  tile {ti}: (gap {gap}) k-loop {int(T[1] - T[0])} | next-tile issue {int(T[2] - T[1])} | epilogue {int(T[3] - T[2])}"
              f" = entry {int(E[0] - T[2])} + bias/table {int(E[1] - E[0])} + slab0 {int(E[2] - E[1])} + slabs1-3 {int(E[3] - E[2])} + exit {int(T[3] - E[3])}")
        prev_t3 = T[3]


B = 32
M = B * 577
bf = torch.bfloat16
for name, N, K, act, inplace32 in (("qkv-like (plain)", 2304, 768, 0, False), ("fc1-like (GELU)", 3072, 768, 2, False),
                                   ("fc2-like (fp32 residual in place)", 768, 3072, 0, True)):
    A = torch.randn(M, K, device="cuda").to(bf); W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(bf)
    C = torch.zeros(M, N, device="cuda", dtype=torch.float32 if inplace32 else bf)
    bias = torch.randn(N, device="cuda")
    R = C.data_ptr() if inplace32 else None
    report(f"{name} M={M} N={N} K={K}", lambda: lib.dptx_op_gemm(0, A.data_ptr(), W.data_ptr(), bias.data_ptr(), R, C.data_ptr(), M, N, K, act, 0,
                                                               int(inplace32), int(inplace32), st))
    del A, W, C
X = torch.randn(B, 96, 96, 256, device="cuda").to(bf)
Wt = (torch.randn(256, 3, 3, 256, device="cuda") * (9 * 256) ** -0.5).to(bf)
Y = torch.empty(B, 96, 96, 256, device="cuda", dtype=bf)
report("rcu@96 (3x3 conv, K = 2304)", lambda: lib.dptx_op_conv(0, X.data_ptr(), Wt.data_ptr(), None, None, Y.data_ptr(), B, 96, 96, 256, 256, 3, 1, 1, 1, 96, 96, 0, 0, st))
