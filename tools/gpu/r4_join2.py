"""Round-4 diagnosis, step 3: r4_join.py saw ONE forward in 2400 (bf16, B = 6, three streams, caller = the NULL stream)
whose comparison on the caller's stream mismatched in 441 840 elements (one image) with no NaN left afterwards.
Unordered join (the comparison read the image before the sub-stream had written it) or a wrong image?

Per forward: NaN-fill a fresh output on the caller's stream, run the forward, SNAPSHOT the output with one copy kernel on the
caller's stream (no host sync), keep going; every `chunk` forwards synchronise and classify each snapshot against the
reference: NaN in the snapshot = the copy ran before the writers (join not ordered); finite-but-different = wrong values.
The live output buffers are re-compared after the synchronisation as well.

  python tools/gpu/r4_join2.py [iters]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omnidata_amd.engine import Engine  # noqa: E402
from omnidata_amd.weights import random_dual_state_dict, random_state_dict, synthetic_input  # noqa: E402

DEV = "cuda:0"


def run(dtype, dual, B, streams, iters, side_stream, chunk=25):
    eng = Engine(num_channels=3, max_batch=B, dtype=dtype, device_id=0, dual=dual, streams=streams)
    eng.load_state_dict(random_dual_state_dict(3) if dual else random_state_dict(3, 3))
    x = synthetic_input(11, B, "normal").to(DEV)
    if dtype == "fp8":
        eng.calibrate_fp8(x)
    st = torch.cuda.Stream() if side_stream else torch.cuda.current_stream()
    stats = {"join": 0, "wrong": 0, "live_wrong": 0}
    with torch.cuda.stream(st):
        ref = torch.empty(B, 3, 384, 384, device=DEV)
        refd = torch.empty(B, 1, 384, 384, device=DEV)
        if dual:
            eng.forward_dual(x, out_normal=ref, out_depth=refd)
        else:
            eng.forward(x, out=ref)
        torch.cuda.synchronize()
        done = 0
        while done < iters:
            n = min(chunk, iters - done)
            outs, snaps = [], []
            for _ in range(n):
                on = torch.full((B, 3, 384, 384), float("nan"), device=DEV)
                od = torch.full((B, 1, 384, 384), float("nan"), device=DEV) if dual else None
                if dual:
                    eng.forward_dual(x, out_normal=on, out_depth=od)
                    snaps.append((on.clone(), od.clone()))
                else:
                    eng.forward(x, out=on)
                    snaps.append((on.clone(), None))
                outs.append((on, od))
            torch.cuda.synchronize()
            for i, ((sn, sd), (on, od)) in enumerate(zip(snaps, outs)):
                for nm, s_, o_, r_ in (("normal", sn, on, ref), ("depth", sd, od, refd)):
                    if s_ is None:
                        continue
                    if not torch.equal(s_, r_):
                        m = s_ != r_
                        nan = int((torch.isnan(s_) & m).sum())
                        per_img = [int(v) for v in m.flatten(1).sum(1)]
                        mx = float((s_ - r_).abs().nan_to_num(0.0).max())
                        kind = "join" if nan > 0 else "wrong"
                        stats[kind] += 1
                        print(f"   forward {done + i} {nm}: snapshot differs in {int(m.sum())} elements per_image={per_img} "
                              f"NaN among them {nan}, max finite |d| {mx:.3e}; live buffer after sync equal: {bool(torch.equal(o_, r_))}", flush=True)
                    if not torch.equal(o_, r_):
                        stats["live_wrong"] += 1
            done += n
    print(f"[join2 {dtype:6s} dual={int(dual)} B={B} streams={streams} caller={'side' if side_stream else 'null'}] {iters} forwards: "
          f"snapshots with NaN (join unordered) {stats['join']}, snapshots finite-but-different {stats['wrong']}, "
          f"live buffers wrong after sync {stats['live_wrong']}", flush=True)
    eng.close()
    return stats


if __name__ == "__main__":
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    for dtype, dual, B, streams, side in (("bf16", False, 6, 3, False), ("bf16", False, 6, 2, False), ("fp8", True, 3, 2, False),
                                          ("bf16", False, 6, 3, True), ("bf16", False, 2, 2, False)):
        run(dtype, dual, B, streams, iters, side)
