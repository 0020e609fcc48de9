"""Round 5 probe: how much does CROSS-FORWARD overlap buy?  The shipped schedule runs the two halves of ONE forward in lockstep
on two streams and joins them before the next forward may start, so the latency-bound ResNetV2 stages of both halves meet each
other, never the MFMA-bound ViT / decoder launches.  Here H engine handles (each with its own arena and internal stream count)
free-run K forwards each on H torch streams with no synchronisation between steps: aggregate images/s.

  python tools/gpu/pipeline_probe.py [--dtype bf16] [--steps 20]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omnidata_amd.engine import Engine  # noqa: E402
from omnidata_amd.weights import random_state_dict, synthetic_input  # noqa: E402


def run(tag, handles, per, inner_streams, dtype, steps, sd, stagger_ms=0.0):
    engs = [Engine(num_channels=3, max_batch=per, dtype=dtype, device_id=0, streams=inner_streams) for _ in range(handles)]
    engs[0].load_state_dict(sd)
    blob = engs[0].export_packed()
    torch.cuda.synchronize()
    for e in engs[1:]:
        e.import_packed(blob)
    xs = [synthetic_input(1000 + i, per, "normal").cuda() for i in range(handles)]
    ys = [torch.empty(per, 3, 384, 384, device="cuda") for _ in range(handles)]
    streams = [torch.cuda.Stream() for _ in range(handles)]

    def loop(n):
        for _ in range(n):
            for e, st, x, y in zip(engs, streams, xs, ys):
                with torch.cuda.stream(st):
                    e.forward(x, out=y)

    loop(3)
    torch.cuda.synchronize()
    if stagger_ms > 0 and handles > 1:   # offset the later handles by a spin on their streams
        for i, st in enumerate(streams[1:], 1):
            with torch.cuda.stream(st):
                torch.cuda._sleep(int(stagger_ms * i * 2.0e6))   # ~cycles at ~2 GHz
    t0 = time.perf_counter()
    loop(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ips = handles * per * steps / dt
    print(f"{tag:58s} {ips:8.1f} img/s   {1e3 * dt / steps:7.3f} ms per round of {handles} x {per}", flush=True)
    for e in engs:
        e.close()
    del engs, xs, ys
    torch.cuda.empty_cache()
    return ips


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    sd = random_state_dict(0, 3)
    for rep in range(2):
        run("1 handle  x 32, inner streams 2 (shipped schedule)", 1, 32, 2, args.dtype, args.steps, sd)
        run("1 handle  x 32, inner streams 1", 1, 32, 1, args.dtype, args.steps, sd)
        run("2 handles x 32, inner streams 1, free-running", 2, 32, 1, args.dtype, args.steps, sd)
        run("2 handles x 32, inner streams 1, staggered 6 ms", 2, 32, 1, args.dtype, args.steps, sd, stagger_ms=6.0)
        run("2 handles x 32, inner streams 2, free-running", 2, 32, 2, args.dtype, args.steps, sd)
        run("3 handles x 32, inner streams 1, free-running", 3, 32, 1, args.dtype, args.steps, sd)
        run("2 handles x 16, inner streams 1, free-running", 2, 16, 1, args.dtype, 2 * args.steps, sd)
        run("2 handles x 16, inner streams 1, staggered 3 ms", 2, 16, 1, args.dtype, 2 * args.steps, sd, stagger_ms=3.0)
        run("4 handles x 16, inner streams 1, free-running", 4, 16, 1, args.dtype, 2 * args.steps, sd)


if __name__ == "__main__":
    main()
