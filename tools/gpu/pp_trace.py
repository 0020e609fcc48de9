"""Slot timing of the ping-pong 256x256 kernel (gemm_pp_kernel): s_memtime stamps of block 0 (experiment build -DDPTX_TRACE).
Stamps per k-tile:  group 0 (waves 0-3): 0 start | 1 DMA issued | 2 past barrier A | 3 MFMAs + vmcnt done
                    group 1 (waves 4-7): 0 start | 1 MFMAs (+ vmcnt) done | 2 past barrier A | 3 DMA issued
DPTX_HALO=0 keeps the convolution on this kernel."""
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
    t = buf.cpu()
    cal = t[0, 62]
    cyc, wall = int(cal[2] - cal[0]), int(cal[3] - cal[1])
    print(f"== {name}" + (f"   [{cyc / (wall * 10.0):.2f} cycles/ns over the k-loop]" if wall > 0 else ""))
    for w in (0, 3, 4, 7):
        tw = t[w]
        n = min(int((tw[:60, 0] != 0).sum()), 60)
        if n < 6:
            print("  wave", w, "no data"); continue
        s = tw[:n].double()
        it = (s[1:, 0] - s[:-1, 0])[2:].mean()
        a = (s[:, 1] - s[:, 0])[2:-1].mean()
        b = (s[:, 2] - s[:, 1])[2:-1].mean()
        c = (s[:, 3] - s[:, 2])[2:-1].mean()
        d = (s[1:, 0] - s[:-1, 3])[2:].mean()
        if os.environ.get("DPTX_PP") == "5":   # single-barrier variant: 0 start | 1 | 2 | 3 before the barrier
            if w < 4:
                print(f"  wave {w} (group 0): iteration {it:.0f} = DMA issue {a:.0f} + MFMAs {b:.0f} + vmcnt(0) {c:.0f} + barrier wait {d:.0f}")
            else:
                print(f"  wave {w} (group 1): iteration {it:.0f} = MFMAs {a:.0f} + DMA issue {b:.0f} + vmcnt(4) {c:.0f} + barrier wait {d:.0f}")
        elif w < 4:
            print(f"  wave {w} (group 0): iteration {it:.0f} = DMA issue {a:.0f} + barrier wait {b:.0f} + MFMA slot (reads, 32 MFMAs, vmcnt) {c:.0f} + barrier wait {d:.0f}")
        else:
            print(f"  wave {w} (group 1): iteration {it:.0f} = MFMA slot (reads, 32 MFMAs, vmcnt) {a:.0f} + barrier wait {b:.0f} + DMA issue {c:.0f} + barrier wait {d:.0f}")


B = 32
X = torch.randn(B, 96, 96, 256, device="cuda").to(torch.bfloat16)
Wt = (torch.randn(256, 3, 3, 256, device="cuda") * (9 * 256) ** -0.5).to(torch.bfloat16)
Y = torch.empty(B, 96, 96, 256, device="cuda", dtype=torch.bfloat16)
report("rcu@96 bf16 (3x3 conv, K = 2304)", lambda: lib.dptx_op_conv(0, X.data_ptr(), Wt.data_ptr(), None, None, Y.data_ptr(), B, 96, 96, 256, 256, 3, 1, 1, 1, 96, 96, 0, 0, st))
del X, Wt, Y
for M, N, K in ((32 * 577, 768, 3072), (8192, 8192, 8192)):
    A = torch.randn(M, K, device="cuda").to(torch.bfloat16); W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    report(f"dense M={M} N={N} K={K}", lambda: lib.dptx_op_gemm(0, A.data_ptr(), W.data_ptr(), None, None, C.data_ptr(), M, N, K, 0, 0, 0, 0, st))
    del A, W, C
