"""Test infrastructure (oracle/): times the REFERENCE's own modules against the CPU port that `bench.py` times.

VERDICT r4 W9 asks for `cpu_baseline.kind = "reference"`.  The reference is Python: it imports HERE (this container holds
/root/reference) but does not travel to the GPU box, where `bench.py` runs -- so the bench line can only time the port
(oracle/dpt_oracle.py, pinned at 0.0 against the reference by oracle/validate_vs_reference.py) and says `kind: "port"`.
This script closes the gap the only way the set-up allows: both forwards on the SAME host, same threads, same seeded
weights and inputs, alternating, and their ratio.  With the ratio within a few per cent, the port's img/s on the GPU box's
host IS the reference's, to that accuracy.

    python -m oracle.time_reference_vs_port [--batch 4] [--reps 3]     (writes nothing; paste into profiles/)
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/omnidata_tools/torch"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    assert os.path.isdir(REF), "reference checkout not present (this runs in the build container only)"
    from oracle import timm_shim
    from oracle.dpt_oracle import dpt_forward
    from omnidata_amd.weights import random_state_dict, synthetic_input
    timm_shim.install()
    sys.path.insert(0, REF)
    from modules.midas.dpt_depth import DPTDepthModel   # the reference's own class, unmodified
    nt = os.cpu_count() or 1
    torch.set_num_threads(nt)
    sd = random_state_dict(0, 3)
    model = DPTDepthModel(backbone="vitb_rn50_384", num_channels=3).eval()
    model.load_state_dict(sd, strict=True)
    x = synthetic_input(1000, args.batch, "normal")
    with torch.no_grad():
        y_ref = model(x)          # warm-up of both
        y_port = dpt_forward(sd, x)
    print(f"host: {nt} logical CPUs, torch {torch.__version__}, batch {args.batch}, fp32; "
          f"max|reference - port| = {float((y_ref - y_port.reshape(y_ref.shape)).abs().max()):.3e}")
    t_ref, t_port = [], []
    for _ in range(args.reps):    # alternating: drift of the host hits both alike
        with torch.no_grad():
            t0 = time.perf_counter(); model(x); t_ref.append(time.perf_counter() - t0)
            t0 = time.perf_counter(); dpt_forward(sd, x); t_port.append(time.perf_counter() - t0)
    med = lambda v: sorted(v)[len(v) // 2]
    r, p = args.batch / med(t_ref), args.batch / med(t_port)
    print(f"reference modules + timm shim : {r:.3f} images/s   (runs: {[round(args.batch / t, 3) for t in t_ref]})")
    print(f"oracle/dpt_oracle.py (port)   : {p:.3f} images/s   (runs: {[round(args.batch / t, 3) for t in t_port]})")
    print(f"port / reference = {p / r:.3f}")


if __name__ == "__main__":
    main()
