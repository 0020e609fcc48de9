"""Which decoder convolutions tolerate e4m3 operands?  (TEST INFRASTRUCTURE ONLY -- CPU emulation on the fp32 oracle.)

BASELINE.json configs[4] asks for "fp8 MFMA weights".  The fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4) takes e4m3 / e5m2 /
fp6 / fp4 on BOTH operands -- there is no e4m3 x bf16 form -- so "fp8 weights at the fp8 rate" means e4m3 activations too,
layer by layer.  Round 3 ran all 19 eligible decoder convolutions (14 RCU 3x3, 4 out_conv, output_conv.0) that way: 9.4 deg /
7.5 deg mean angular error against 4.2 deg / 1.1 deg of the bf16 engine on the two synthetic weight families.

This script emulates the engine's quantisation -- weights per OUTPUT CHANNEL after a power-of-two scale into (224, 448],
activations per tensor after a power-of-two scale that puts the tensor's max |x| into (112, 224] (ReLU'd first where the
consumer pre-activates), both rounded to OCP e4m3 with saturation -- for ONE layer at a time on the fp32 oracle, records the
rms / mean angular error it causes, and then evaluates layer SETS (errors add in variance to first order; the sets are
verified with one emulated forward each).

  python -m oracle.fp8_layers [--out profiles/r04_fp8_layers.json]
"""
import argparse
import json
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle.dpt_oracle as O  # noqa: E402
from omnidata_amd.weights import random_state_dict, synthetic_input  # noqa: E402

E4M3 = torch.float8_e4m3fn


def q_e4m3(t):
    return t.clamp(-448.0, 448.0).to(E4M3).float()


def quant_weight(w):
    mx = w.abs().flatten(1).amax(1).clamp_min(1e-30)
    k = torch.floor(torch.log2(448.0 / mx))
    sc = torch.pow(2.0, k).view(-1, 1, 1, 1)
    return q_e4m3(w * sc) / sc


def quant_act(a, relu):
    x = F.relu(a) if relu else a
    mx = float(x.abs().max())
    k = math.floor(math.log2(224.0 / mx)) if mx > 0 else 0
    sc = 2.0 ** k
    return q_e4m3(x * sc) / sc


FP8_LAYERS = []
for i in (4, 3, 2, 1):
    for u in (1, 2):
        if i == 4 and u == 1:
            continue
        for c in (1, 2):
            FP8_LAYERS.append(f"scratch.refinenet{i}.resConfUnit{u}.conv{c}.weight")
    FP8_LAYERS.append(f"scratch.refinenet{i}.out_conv.weight")
FP8_LAYERS.append("scratch.output_conv.0.weight")


def run(sd, x, layers):
    """fp32 oracle forward with the convolutions in `layers` on e4m3 operands."""
    ids = {id(v): k for k, v in sd.items()}
    conv0 = F.conv2d

    def conv(a, w, b=None, *aa, **kw):
        k = ids.get(id(w))
        if k in layers:
            # RCU conv1 consumes relu(x) (the oracle applies the ReLU before calling conv: `a` is already activated);
            # conv2 consumes relu(conv1) likewise; out_conv / output_conv.0 consume the tensor as it is
            return conv0(quant_act(a, False), quant_weight(w), b, *aa, **kw)
        return conv0(a, w, b, *aa, **kw)

    class Fp:
        def __getattr__(self, n):
            return conv if n == "conv2d" else getattr(F, n)

    old = O.F
    O.F = Fp()
    try:
        return O.dpt_forward(sd, x)
    finally:
        O.F = old


def ang(y, ref):
    return O.mean_angular_error_deg(y.clamp(0, 1), ref.clamp(0, 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    O.oracle_threads()
    res = {}
    for fam in ("default", "trained"):
        sd = random_state_dict(a.seed, 3, family=fam)
        x = synthetic_input(a.seed, 1, "normal")
        ref = O.dpt_forward(sd, x)
        per = {}
        for k in FP8_LAYERS:
            y = run(sd, x, {k})
            per[k] = {"rms": float((y - ref).pow(2).mean().sqrt()), "ang": float(ang(y, ref))}
            print(f"[{fam}] {k:58s} rms {per[k]['rms']:.3e}  ang {per[k]['ang']:.3f} deg", flush=True)
        sets = {
            "all 19 (round 3)": set(FP8_LAYERS),
            "refinenet1 RCUs": {k for k in FP8_LAYERS if "refinenet1.resConfUnit" in k},
            "refinenet1+2 RCUs": {k for k in FP8_LAYERS if ("refinenet1.resConfUnit" in k or "refinenet2.resConfUnit" in k)},
            "all RCUs of refinenet1..3": {k for k in FP8_LAYERS if "resConfUnit" in k and "refinenet4" not in k},
            "refinenet1+2 RCUs + output_conv.0": {k for k in FP8_LAYERS if ("refinenet1.resConfUnit" in k or "refinenet2.resConfUnit" in k or "output_conv.0" in k)},
        }
        ev = {}
        for name, S in sets.items():
            y = run(sd, x, S)
            pred = math.sqrt(sum(per[k]["rms"] ** 2 for k in S))
            ev[name] = {"rms": float((y - ref).pow(2).mean().sqrt()), "ang": float(ang(y, ref)), "predicted_rms": pred, "layers": sorted(S)}
            print(f"[{fam}] SET {name:38s} rms {ev[name]['rms']:.3e} (sum of parts {pred:.3e})  ang {ev[name]['ang']:.3f} deg", flush=True)
        res[fam] = {"per_layer": per, "sets": ev}
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
