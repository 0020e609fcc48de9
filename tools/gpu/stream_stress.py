"""Stress check of the two-stream schedule: N forwards of the same batch must all equal the single-stream result bit for bit.
Usage (GPU box): python tools/gpu/stream_stress.py 200 [streams]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omnidata_amd.engine import Engine
from omnidata_amd.weights import random_state_dict, synthetic_input
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sd = random_state_dict(0, 3)
x = synthetic_input(5, 32, "normal").to("cuda:0")
ref_eng = Engine(num_channels=3, max_batch=32, dtype="bf16", device_id=0, streams=1); ref_eng.load_state_dict(sd)
ref = ref_eng.forward(x).clone()
torch.cuda.synchronize()
eng = Engine(num_channels=3, max_batch=32, dtype="bf16", device_id=0, streams=NS); eng.load_state_dict(sd)
bad = 0; which = []
N = int(sys.argv[1])
for k in range(N):
    y = eng.forward(x)
    torch.cuda.synchronize()
    d = (y - ref).abs().flatten(1).max(1).values
    b = [i for i, v in enumerate(d) if v > 0]
    if b:
        bad += 1; which.append((k, b))
print(os.environ.get("TAG", ""), NS, "streams; bad runs:", bad, "of", N, which[:6])
