"""Round 6, same-process alternating A/B of the image-group split of the ResNetV2 section (engine.hip dptx_engine::n_groups,
DPTX_STAGE_GROUPS is read by dptx_create) under the three schedules: one stream, two half-batches, two forwards in flight.
Every configuration's output is compared bit for bit with the plain single-stream forward.
    python tools/gpu/r6_groups_ab.py [--dtype bf16] [--reps 3] [--steps 16]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omnidata_amd.engine import Engine  # noqa: E402
from omnidata_amd.pipeline import ForwardPipeline  # noqa: E402
from omnidata_amd.weights import random_state_dict, synthetic_input  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--groups", default="1,2,4,8")
    args = ap.parse_args()
    B = args.batch
    os.environ["DPTX_STAGE_GROUPS"] = "1"
    owner = Engine(num_channels=3, max_batch=B, dtype=args.dtype, device_id=0, streams=1)
    owner.load_state_dict(random_state_dict(0, 3))
    x = synthetic_input(1000, B, "normal").cuda()
    ref = owner.forward(x).clone()
    torch.cuda.synchronize()
    groups = [int(g) for g in args.groups.split(",")]
    cfgs = [(s, g, f) for f, s in ((1, 1), (1, 2), (2, 1)) for g in groups]
    res = {c: [] for c in cfgs}
    for rep in range(args.reps):
        for (s, g, f) in cfgs:
            os.environ["DPTX_STAGE_GROUPS"] = str(g)
            if f == 1:
                e = Engine(num_channels=3, max_batch=B, dtype=args.dtype, device_id=0, streams=s)
                e.share_weights_from(owner)
                ys = [torch.empty_like(ref)]
                run = lambda i: e.forward(x, out=ys[0])
                close = e.close
            else:
                pipe = ForwardPipeline.from_engine(owner, depth=f)
                ys = [torch.empty_like(ref) for _ in range(f)]
                run = lambda i: pipe.submit(x, out=ys[i % f])
                close = pipe.close
            for i in range(4):
                run(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                run(i)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            same = all(torch.equal(y, ref) for y in ys)
            res[(s, g, f)].append(B * args.steps / dt)
            print(f"rep {rep} streams={s} groups={g} inflight={f}: {B * args.steps / dt:8.1f} img/s  bit-identical={same}", flush=True)
            close()
    print(f"\n== {args.dtype} B={B} median of {args.reps}")
    for c in cfgs:
        v = sorted(res[c])
        print(f"streams={c[0]} groups={c[1]} inflight={c[2]}: median {v[len(v) // 2]:8.1f}  all {[round(a, 1) for a in res[c]]}")


if __name__ == "__main__":
    main()
