"""In-network: where does the forward with the register-direct GEMM epilogue first differ (bit for bit) from the forward
with the staged epilogue (dptx_debug_set_gemm_flags(1))?  B = 32, bf16, stage taps."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omnidata_amd.model import DPTDepthModel
from omnidata_amd.weights import random_state_dict
from omnidata_amd.engine import load_library

lib = load_library()
B = int(os.environ.get("PROBE_B", "32"))
model = DPTDepthModel(num_channels=3, dtype="bf16", max_batch=B)
model.load_state_dict(random_state_dict(0, 3))
model.to("cuda:0")
x = torch.rand(B, 3, 384, 384, generator=torch.Generator().manual_seed(5)).cuda()
names = ["stem", "s0", "s1", "s2", "tok0", "blk0", "blk3", "blk8", "blk11", "l3", "l4", "l1_rn", "l2_rn", "l3_rn", "l4_rn", "p4", "p3",
         "p2", "p1", "h0", "h1"]
res = {}
for flags in (0, 1):
    lib.dptx_debug_set_gemm_flags(flags)
    eng = model._get_engine(torch.device("cuda:0"))
    eng.enable_taps(True)
    y = model(x).clone()
    taps = {n: eng.tap(n).clone() for n in names}
    eng.enable_taps(False)
    y2 = model(x).clone()   # the product's schedule: two streams, fused head tail
    res[flags] = (taps, y2)
lib.dptx_debug_set_gemm_flags(0)
for n in names:
    a, b = res[0][0][n], res[1][0][n]
    nd = (a != b).sum().item()
    print(f"tap {n:6s}: {nd} of {a.numel()} differ" + (f"  max|d| {(a - b).abs().max().item():.3e}" if nd else ""))
print("output:", (res[0][1] != res[1][1]).sum().item(), "differ")
# batch invariance of each form
for flags in (0, 1):
    lib.dptx_debug_set_gemm_flags(flags)
    y1 = model(x[:1])
    print(f"flags={flags}: image 0 alone equals image 0 of the batch: {torch.equal(y1[0], res[flags][1][0])}"
          f"  (max|d| {(y1[0] - res[flags][1][0]).abs().max().item():.3e})")
lib.dptx_debug_set_gemm_flags(0)
