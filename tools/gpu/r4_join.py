"""Round-4 diagnosis, second hypothesis for the fp8 dual-task "nondeterminism" (VERDICT r3 W1): the forward's JOIN.

A forward of >= 2 images runs on internal streams and is joined to the caller's stream with events.  The round-3 stress test
compared every forward with the reference on the caller's stream, but into ONE output buffer that already held the identical
result of the previous forward -- a comparison that ran BEFORE the sub-streams had finished would have read old, identical
bytes and passed.  The failing driver test compares two FRESH torch.empty() outputs right after the second forward.

Here every forward writes into a buffer that was NaN-filled on the caller's stream just before, and the comparison follows
on the caller's stream with no host synchronisation in between: an unordered join shows as NaN / mismatching elements.

  python tools/gpu/r4_join.py [iters]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omnidata_amd.engine import Engine  # noqa: E402
from omnidata_amd.weights import random_dual_state_dict, random_state_dict, synthetic_input  # noqa: E402

DEV = "cuda:0"


def run(dtype, dual, B, streams, iters, side_stream):
    eng = Engine(num_channels=3, max_batch=B, dtype=dtype, device_id=0, dual=dual, streams=streams)
    eng.load_state_dict(random_dual_state_dict(3) if dual else random_state_dict(3, 3))
    x = synthetic_input(11, B, "normal").to(DEV)
    if dtype == "fp8":
        eng.calibrate_fp8(x)
    st = torch.cuda.Stream() if side_stream else torch.cuda.current_stream()
    with torch.cuda.stream(st):
        def fwd(out, out2):
            if dual:
                eng.forward_dual(x, out_normal=out, out_depth=out2)
            else:
                eng.forward(x, out=out)
        ref = torch.empty(B, 3, 384, 384, device=DEV)
        ref2 = torch.empty(B, 1, 384, 384, device=DEV)
        fwd(ref, ref2)
        torch.cuda.synchronize()
        bad = torch.zeros((), dtype=torch.int64, device=DEV)
        nan = torch.zeros((), dtype=torch.int64, device=DEV)
        bad_iters = torch.zeros((), dtype=torch.int64, device=DEV)
        for _ in range(iters):
            out = torch.full((B, 3, 384, 384), float("nan"), device=DEV)
            out2 = torch.full((B, 1, 384, 384), float("nan"), device=DEV)
            fwd(out, out2)
            d = (out != ref).sum() + ((out2 != ref2).sum() if dual else 0)
            bad += d
            bad_iters += (d > 0).long()
            nan += torch.isnan(out).sum()
        torch.cuda.synchronize()
    print(f"[join {dtype:6s} dual={int(dual)} B={B} streams={streams} caller={'side' if side_stream else 'null'} stream] "
          f"{iters} forwards: {int(bad_iters)} with mismatches, {int(bad)} elements, {int(nan)} still NaN", flush=True)
    eng.close()
    return int(bad_iters)


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    total = 0
    for dtype, dual, B, streams in (("fp8", True, 3, 2), ("bf16", True, 3, 2), ("bf16", False, 2, 2), ("bf16", False, 6, 3), ("mixed", False, 3, 2),
                                    ("bf16", False, 3, 1)):
        for side in (False, True):
            total += run(dtype, dual, B, streams, iters, side)
    print(f"TOTAL forwards with mismatches: {total}")
