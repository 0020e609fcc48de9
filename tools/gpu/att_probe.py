"""Where does an attention launch's time go?  Runs tools/gpu/probes/libattprobe.so (a copy of csrc/attention.hip with an
ablation mask, see its header) on one block's qkv tensor (B = 32, S = 577, 12 heads, bf16) next to the library's kernel.
  python tools/gpu/att_probe.py
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from omnidata_amd.engine import DTYPES, load_library  # noqa: E402

lib = load_library()
probe = C.CDLL(os.path.join(ROOT, "tools", "gpu", "probes", "libattprobe.so"))
probe.att_probe.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
B, S, H = 32, 577, 12
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * S, 3 * H * 64, generator=g) * 1.5).to(torch.bfloat16).cuda()
ref = torch.empty(B * S, H * 64, dtype=torch.bfloat16, device="cuda")
out = torch.empty_like(ref)
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


t_lib = timeit(lambda: lib.dptx_op_attention(DTYPES["bf16"], qkv.data_ptr(), ref.data_ptr(), B, S, H, st))
print(f"library attention_kernel: {t_lib:.1f} us per launch")
NAMES = {0: "probe copy, nothing removed", 64: "PACKED softmax arithmetic (correct results)", 1: "no exponential", 2: "no running sum",
         3: "no exponential, no sum", 4: "no running max", 7: "no exponential / sum / max (softmax = one fma + pack)", 8: "no PV MFMAs",
         16: "no QK MFMAs", 24: "no MFMAs at all", 31: "staging + LDS reads + packing only", 32: "no K / V staging after tile 0",
         63: "skeleton: loop, barrier, fragment reads, packing"}
NAMES[128] = "K tile by LDS-DMA (correct results)"; NAMES[192] = "K by LDS-DMA + packed softmax arithmetic (correct results)"
NAMES[256] = "K and V by LDS-DMA, transposing fragment reads (correct results)"; NAMES[320] = "the same + packed softmax arithmetic (correct results)"
for mask in (0, 64, 128, 192, 256, 320, 1, 2, 3, 4, 7, 8, 16, 24, 32, 31, 63):
    out.zero_()
    rc = probe.att_probe(mask, qkv.data_ptr(), out.data_ptr(), B, S, H, st)
    assert rc == 0, rc
    torch.cuda.synchronize()
    note = ""
    if mask in (0, 64, 128, 192, 256, 320):
        d = (out.float() - ref.float()).abs().max().item()
        note = f"   max |d| vs the library kernel {d:.2e}" + ("  (bit-identical)" if torch.equal(out, ref) else "")
    t = timeit(lambda: probe.att_probe(mask, qkv.data_ptr(), out.data_ptr(), B, S, H, st))
    print(f"mask {mask:2d}  {t:6.1f} us   {NAMES[mask]}{note}")

t_lib2 = timeit(lambda: lib.dptx_op_attention(DTYPES["bf16"], qkv.data_ptr(), ref.data_ptr(), B, S, H, st))
print(f"library attention_kernel, measured again at the end: {t_lib2:.1f} us per launch")
