"""A/B process of profiles/r06_experiments.md section 7: one seeded B = 32 forward (SHA of the output) + 3 x 20 timed forwards on the
single-stream schedule.  `DPTX_STREAMS=1 DPTX_SIDE={0,1} python tools/gpu/side_ab.py {bf16,mixed}`; DPTX_SIDE is read by the library
only with profiles/r06_side_stream_attempt.patch applied (the experiment was not kept)."""
import hashlib, os, sys, time, torch
sys.path.insert(0, os.getcwd())
from omnidata_amd.model import build_model
torch.manual_seed(0)
dt = sys.argv[1]
m = build_model("normal", dtype=dt, max_batch=32, random_weights=3)
dev = torch.device("cuda:0")
m.to(dev)
x = torch.rand(32, 3, 384, 384, device=dev)
y = m(x)
torch.cuda.synchronize()
h = hashlib.sha256(y.float().cpu().numpy().tobytes()).hexdigest()[:16]
for _ in range(5): m(x)
torch.cuda.synchronize()
ts = []
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20): m(x)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) / 20 * 1e3)
print(f"SIDE={os.environ.get('DPTX_SIDE','0')} dtype={dt} sha={h} ms/forward={min(ts):.3f} ({' '.join('%.3f'%t for t in ts)}) img/s={32/min(ts)*1e3:.0f}", flush=True)
