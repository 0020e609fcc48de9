"""The software-pipelined attention prototype (tools/gpu/probes/attention_pipe_probe.hip: QK^T of tile t + 1 before the softmax of
tile t) next to the library's kernel: results and microseconds per launch.  python tools/gpu/att_pipe_probe.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from omnidata_amd.engine import DTYPES, load_library  # noqa: E402

lib = load_library()
B, S, H = 32, 577, 12
g = torch.Generator().manual_seed(0)
qkv = (torch.randn(B * S, 3 * H * 64, generator=g) * 1.5).to(torch.bfloat16).cuda()
ref = torch.empty(B * S, H * 64, dtype=torch.bfloat16, device="cuda")
out = torch.empty_like(ref)
st = torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


run_lib = lambda: lib.dptx_op_attention(DTYPES["bf16"], qkv.data_ptr(), ref.data_ptr(), B, S, H, st)
timeit(run_lib)   # clock ramp
print(f"library attention_kernel: {timeit(run_lib):.1f} us per launch")
for blocks in (2, 3):
    p = C.CDLL(os.path.join(ROOT, "tools", "gpu", "probes", f"libattpipe{blocks}.so"))
    p.att_pipe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    out.zero_()
    assert p.att_pipe(qkv.data_ptr(), out.data_ptr(), B, S, H, st) == 0
    torch.cuda.synchronize()
    d = (out.float() - ref.float()).abs().max().item()
    t = timeit(lambda: p.att_pipe(qkv.data_ptr(), out.data_ptr(), B, S, H, st))
    print(f"pipelined prototype, {blocks} blocks per CU: {t:.1f} us per launch; max |d| vs the library kernel {d:.2e}" + ("  (bit-identical)" if torch.equal(out, ref) else ""))
print(f"library attention_kernel again: {timeit(run_lib):.1f} us per launch")
