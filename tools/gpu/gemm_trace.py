"""Where does a k-loop iteration of the GEMM kernel spend its time?  s_memtime stamps of block 0 (dptx_debug_set_trace)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omnidata_amd.engine import load_library

lib = load_library()
st = torch.cuda.current_stream().cuda_stream
buf = torch.zeros(8, 64, 4, dtype=torch.int64, device="cuda")

def report(name, fn, waves):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf.zero_()
    lib.dptx_debug_set_trace(buf.data_ptr())
    fn()
    torch.cuda.synchronize()
    lib.dptx_debug_set_trace(None)
    t = buf.cpu()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    cal = t[0, 62]
    cyc, wall = int(cal[2] - cal[0]), int(cal[3] - cal[1])
    if wall > 0:
        print(f"   block 0 k-loop: {cyc} s_memtime ticks in {wall} wall_clock64 ticks (100 MHz) -> {cyc / (wall * 10.0):.2f} ticks/ns;"
              f" launch {e0.elapsed_time(e1) * 100:.1f} us")
    print(f"== {name}: per k-tile cycles (s_memtime, 100 MHz*? units as read), waves 0..{waves - 1}")
    for w in (0, waves // 2, waves - 1):
        tw = t[w]
        n = int((tw[:, 0] != 0).sum())
        if n < 4:
            print("  wave", w, "no data"); continue
        it = (tw[1:n, 0] - tw[:n - 1, 0]).float()
        issue = (tw[:n, 1] - tw[:n, 0]).float()
        mma = (tw[:n, 2] - tw[:n, 1]).float()
        wait = (tw[1:n, 3] - tw[:n - 1, 2]).float()
        bar = (tw[1:n, 0] - tw[1:n, 3]).float()
        print(f"  wave {w}: iterations {n}; iteration {it[2:].mean():.0f} = DMA issue {issue[2:-1].mean():.0f} + reads/MFMA {mma[2:-1].mean():.0f}"
              f" + vmcnt wait {wait[2:].mean():.0f} + barrier {bar[2:].mean():.0f}   (first its: {[int(x) for x in it[:4]]})")

B = 32
for name, H, Cin, Cout in (("rcu@96 bf16 (256x256 tile)", 96, 256, 256), ("head.0 bf16 (128x128 tile)", 192, 256, 128)):
    X = torch.randn(B, H, H, Cin, device="cuda").to(torch.bfloat16)
    Wt = (torch.randn(Cout, 3, 3, Cin, device="cuda") * (9 * Cin) ** -0.5).to(torch.bfloat16)
    Y = torch.empty(B, H, H, Cout, device="cuda", dtype=torch.bfloat16)
    report(name, lambda: lib.dptx_op_conv(0, X.data_ptr(), Wt.data_ptr(), None, None, Y.data_ptr(), B, H, H, Cin, Cout, 3, 1, 1, 1, H, H, 0, 0, st),
           8 if Cout == 256 else 4)
    del X, Wt, Y
X8 = torch.randn(B, 96, 96, 256, device="cuda").to(torch.float8_e4m3fn)
W8 = (torch.randn(256, 3, 3, 256, device="cuda") * 16).to(torch.float8_e4m3fn)
Y = torch.empty(B, 96, 96, 256, device="cuda", dtype=torch.bfloat16)
report("rcu@96 fp8 (256x256 tile)", lambda: lib.dptx_op_conv_fp8(X8.data_ptr(), W8.data_ptr(), None, None, Y.data_ptr(), None, B, 96, 96, 256, 256, 3, 1, 1, 1,
                                                               96, 96, 0, 0, 1.0 / 768, st), 8)
M, N, K = 32 * 577, 3072, 768
A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
report("fc1 bf16 (128x128 tile)", lambda: lib.dptx_op_gemm(0, A.data_ptr(), W.data_ptr(), None, None, C.data_ptr(), M, N, K, 0, 0, 0, 0, st), 4)
