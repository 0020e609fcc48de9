"""Experiment: the batch of 32 split over S engine handles, each on its own HIP stream, so that MFMA-bound and
HBM-bound launches of different sub-batches overlap.  Usage: python tools/multistream_bench.py [--streams 1,2,4]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from omnidata_amd.build import build  # noqa: E402
from omnidata_amd.engine import Engine  # noqa: E402
from omnidata_amd.weights import random_state_dict, synthetic_input  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", default="1,2,4")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--dtype", default="bf16")
    args = ap.parse_args()
    build()
    sd = random_state_dict(0, 3)
    x = synthetic_input(1000, args.batch, "normal").cuda()
    ref = None
    for S in [int(s) for s in args.streams.split(",")]:
        per = args.batch // S
        engs = [Engine(num_channels=3, max_batch=per, dtype=args.dtype, device_id=0) for _ in range(S)]
        engs[0].load_state_dict(sd)
        blob = engs[0].export_packed()
        torch.cuda.synchronize()
        for e in engs[1:]:
            e.import_packed(blob)
        streams = [torch.cuda.Stream() for _ in range(S)]
        y = torch.empty(args.batch, 3, 384, 384, device="cuda")

        def step():
            for i, (e, st) in enumerate(zip(engs, streams)):
                with torch.cuda.stream(st):
                    e.forward(x[i * per:(i + 1) * per], out=y[i * per:(i + 1) * per])

        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        if ref is None:
            ref = y.clone()
        print(f"streams={S} per-stream batch={per}: {dt * 1e3:.3f} ms/step  {args.batch / dt:.1f} img/s  bit-identical={torch.equal(ref, y)}")
        del engs


if __name__ == "__main__":
    main()
