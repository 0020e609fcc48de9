"""Times head_tail_x3_kernel alone (B = 32, 192 x 192 -> 384 x 384) through dptx_op_head_tail; DPTX_HX_DBG ablates phases."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omnidata_amd.engine import load_library
from tests.gpu_util import PlaneArena, ptr, stream
lib = load_library()
B, Hs, Ws, C = 32, 192, 192, 3
ar = PlaneArena(B * Hs * Ws * 128 + 32 * 9 * 128 + 8192, dtype=torch.float16)
g = torch.Generator().manual_seed(0)
H0 = ar.put(torch.randn(B, Hs, Ws, 128, generator=g))
W2 = ar.put(torch.randn(32, 3, 3, 128, generator=g) * 1152 ** -0.5)
b2 = torch.randn(32, device="cuda:0"); w4 = torch.randn(C, 32, device="cuda:0"); b4 = torch.randn(C, device="cuda:0")
y = torch.empty(B, C, 2 * Hs, 2 * Ws, device="cuda:0")
def run():
    assert lib.dptx_op_head_tail(3, ptr(H0), ptr(W2), ptr(b2), ptr(w4), ptr(b4), ptr(y), B, Hs, Ws, C, 1, stream()) == 0
for _ in range(3): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): run()
e1.record(); torch.cuda.synchronize()
print(f"DPTX_HX_DBG={os.environ.get('DPTX_HX_DBG', '0')}: {e0.elapsed_time(e1) / 20:.3f} ms per launch")
