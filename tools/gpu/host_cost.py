"""Host-side cost of enqueuing one forward (time inside dptx_forward, which is asynchronous).  Usage: python tools/gpu/host_cost.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omnidata_amd.engine import Engine
from omnidata_amd.weights import random_state_dict, synthetic_input
sd = random_state_dict(0, 3)
x = synthetic_input(5, 32, "normal").to("cuda:0")
for ns in (1, 2, 3):
    eng = Engine(num_channels=3, max_batch=32, dtype="bf16", device_id=0, streams=ns); eng.load_state_dict(sd)
    y = torch.empty(32, 3, 384, 384, device="cuda:0")
    for _ in range(3):
        eng.forward(x, out=y)
    torch.cuda.synchronize()
    host = []
    t_all = time.perf_counter()
    for _ in range(10):
        t0 = time.perf_counter(); eng.forward(x, out=y); host.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t_all) / 10
    n = eng.info()[0]
    print(f"streams={ns}: host enqueue {1e3 * sum(host) / 10:.2f} ms per forward ({n} launches, {1e6 * sum(host) / 10 / n:.1f} us each); wall {1e3 * t_all:.2f} ms per forward")
    eng.close()
