#!/usr/bin/env python3
"""Error / throughput frontier of the engine's precision modes on an MI355X (profiles/r0N_precision_frontier.md).

For both synthetic weight families (omnidata_amd.weights: 'default' = chaotic random residual net, 'trained' =
trained-like conditioning) and every mode / per-group policy: max|d| and rms of the engine's output against the fp32 CPU
oracle on one image (normal head), and images/s at batch 32 (same protocol as bench.py, fewer steps).

    python tools/precision_frontier.py [--steps 10] [--out gpurun_out/frontier.md]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from omnidata_amd.engine import Engine  # noqa: E402
from omnidata_amd.weights import random_state_dict, synthetic_input  # noqa: E402
from oracle.dpt_oracle import dpt_forward, mean_angular_error_deg, oracle_threads  # noqa: E402

# (dtype, x3 groups, flags, per-layer overrides {conv key: mfmas})
HEAD0 = "scratch.output_conv.0.weight"
MODES = [("bf16", 0, 0, {}), ("fp16", 0, 0, {}), ("mixed", "resnet", 0, {}), ("mixed", "resnet+embed+reassemble+rn+fusion", 0, {}),
         ("mixed", 0, 0, {HEAD0: 1}),   # default per-layer table with the first head conv single-pass too ("policy C")
         ("mixed", 0, 0, {}),           # default: per-layer table of the decoder ("policy A")
         ("mixed", 0, 2, {}),           # DPTX_FLAG_GROUP_POLICY: round 2's default (every group but the ViT blocks)
         ("fp16x3", 0, 0, {}), ("bf16x3", 0, 0, {})]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--out", default=None)
    ap.add_argument("--families", nargs="+", default=["default", "trained"])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    oracle_threads()
    lines = ["| weights | dtype | 3-MFMA layers | max abs d (1 img) | max abs d (4 img) | rms d | mean angular err (deg) | img/s (B=%d) | meets 1e-3 |" % args.batch,
             "|---|---|---|---|---|---|---|---|---|"]
    xb = synthetic_input(1000, args.batch, "normal").to(dev)
    yb = torch.empty(args.batch, 3, 384, 384, device=dev)
    for fam in args.families:
        sd = random_state_dict(0, 3, family=fam)
        x1 = synthetic_input(0, 1, "normal")
        ref = dpt_forward(sd, x1)
        x4 = synthetic_input(1000, 4, "normal")
        ref4 = dpt_forward(sd, x4)
        for dtype, groups, flags, over in MODES:
            eng = Engine(num_channels=3, max_batch=args.batch, dtype=dtype, device_id=0, x3_groups=groups, flags=flags)
            eng.load_state_dict(sd)
            for k, m in over.items():
                eng.set_layer_precision(k, m)
            d4 = (eng.forward(x4.to(dev)).cpu() - ref4).abs().max().item()
            y = eng.forward(x1.to(dev)).cpu()
            d = (y - ref).abs()
            ang = mean_angular_error_deg(y.clamp(0, 1), ref.clamp(0, 1))
            for _ in range(3):
                eng.forward(xb, out=yb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                eng.forward(xb, out=yb)
            torch.cuda.synchronize()
            ips = args.batch * args.steps / (time.perf_counter() - t0)
            g = groups if groups else ("all but vit" if dtype == "mixed" else "-")
            if dtype == "mixed" and not groups:
                g = "all but vit, group level (r2)" if flags else ("per-layer table" + (" + head.0 single" if over else ""))
            line = (f"| {fam} | {dtype} | {g} | {d.max():.2e} | {d4:.2e} | {d.pow(2).mean().sqrt():.2e} | {ang:.3f} | {ips:.0f} | "
                    f"{'yes' if max(d.max().item(), d4) < 1e-3 else 'no'} |")
            print(line, flush=True)
            lines.append(line)
            eng.close()
    if args.out:
        with open(args.out, "w") as f:
            f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
