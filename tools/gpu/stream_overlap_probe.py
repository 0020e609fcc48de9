"""Round 5 probe: do the two streams of the k-th ForwardPipeline created in one process overlap?  (bench.py creates eight or
nine pipelines in a row; the eighth measured 15 % low in three full runs.)  For each of N pipelines created one after the other
from one engine: a two-kernel sleep test on its streams (pair time / single time: 1.0 = concurrent, 2.0 = one after the other)
and its throughput over 12 forwards.

  python tools/gpu/stream_overlap_probe.py [N]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omnidata_amd.engine import Engine  # noqa: E402
from omnidata_amd.pipeline import ForwardPipeline  # noqa: E402
from omnidata_amd.weights import random_state_dict, synthetic_input  # noqa: E402


def sleep_ratio(s0, s1, cycles=4_000_000):
    def run(streams):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for s in streams:
            with torch.cuda.stream(s):
                torch.cuda._sleep(cycles)
        torch.cuda.synchronize()
        return time.perf_counter() - t
    run([s0, s1])
    single = min(run([s0]), run([s1]))
    pair = min(run([s0, s1]), run([s0, s1]))
    return pair / single


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    eng = Engine(num_channels=3, max_batch=32, dtype="bf16", device_id=0)
    eng.load_state_dict(random_state_dict(0, 3))
    x = synthetic_input(1000, 32, "normal").cuda()
    ys = [torch.empty(32, 3, 384, 384, device="cuda") for _ in range(2)]
    for k in range(n):
        pipe = ForwardPipeline.from_engine(eng, depth=2)
        r = sleep_ratio(*pipe.streams)
        for i in range(4):
            pipe.submit(x, out=ys[i % 2])
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(12):
            pipe.submit(x, out=ys[i % 2])
        torch.cuda.synchronize()
        ips = 32 * 12 / (time.perf_counter() - t)
        print(f"pipeline {k:2d}: streams {[int(s.cuda_stream) % 100000 for s in pipe.streams]} priorities {[s.priority for s in pipe.streams]} "
              f"sleep pair/single {r:.2f}   {ips:7.1f} img/s", flush=True)
        pipe.close()


if __name__ == "__main__":
    main()
