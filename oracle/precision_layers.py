"""Per-LAYER precision study, emulated on the CPU oracle.  (TEST INFRASTRUCTURE ONLY)

oracle/precision_policy.py decides per layer GROUP; this script measures every contraction of the
forward on its own.  For layer L and operand o in {a (activation), w (weight)} it rounds ONLY that
operand of ONLY that layer to fp16 (everything else fp32) and records the rms of the output
deviation s(L, o).  Rounding errors of different sites are independent to first order, so the
variance of a whole policy is the sum of the s^2 of the operands it leaves rounded:

    1 MFMA  (a_hi w_hi)                      leaves  s(L,a)^2 + s(L,w)^2
    2 MFMA  (a_hi w_hi + a_hi w_lo)   "w2"    leaves  s(L,a)^2
    2 MFMA  (a_hi w_hi + a_lo w_hi)   "a2"    leaves  s(L,w)^2
    3 MFMA                            "x3"    leaves  ~0

and its cost is MACs(L) x {1, 2, 2, 3}.  `--solve BUDGET` picks, greedily by variance removed per
extra MAC, the cheapest per-layer assignment whose predicted rms is below BUDGET, then VERIFIES it
with one emulated forward in which all the chosen roundings are applied together.

The attention matmuls (QK^T, PV) are one "layer" per block: their operands are q, k, v as stored
by the qkv GEMM and the probabilities.

Usage: python -m oracle.precision_layers [--family default trained] [--seed 0] [--out FILE.json]
                                         [--solve 1.0e-4] [--load FILE.json]
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle.dpt_oracle as O  # noqa: E402
from omnidata_amd.weights import random_state_dict, synthetic_input  # noqa: E402

R16 = lambda t: t.to(torch.float16).float()  # noqa: E731


def run(sd, x, plan):
    """plan: {layer_key: set of operands rounded to fp16, subset of {'a', 'w'}}; attention layers use key
    'pretrained.model.blocks.N.attn' with 'a' = (q, k, v, P)."""
    ids = {id(v): k for k, v in sd.items()}
    ctx = {"key": None}
    macs = {}

    def key_of(w):
        k = ids.get(id(w))
        return k if k is not None else ctx["key"]

    conv0, lin0 = F.conv2d, F.linear

    def conv(a, w, b=None, *aa, **kw):
        k = key_of(w)
        r = plan.get(k, ())
        y = conv0(R16(a) if "a" in r else a, R16(w) if "w" in r else w, b, *aa, **kw)
        macs[k] = float(y.shape[2] * y.shape[3] * w.shape[0] * w.shape[1] * w.shape[2] * w.shape[3])
        return y

    def lin(a, w, b=None):
        k = key_of(w)
        r = plan.get(k, ())
        y = lin0(R16(a) if "a" in r else a, R16(w) if "w" in r else w, b)
        macs[k] = float(y.shape[-2] * w.shape[0] * w.shape[1])
        return y

    class Fp:
        def __getattr__(self, n):
            return {"conv2d": conv, "linear": lin}.get(n, getattr(F, n))

    std0 = O.std_conv_same

    def std_conv(xx, w, stride, ws_eps, ws_form):
        ctx["key"] = ids[id(w)]
        return std0(xx, w, stride, ws_eps, ws_form)

    old_block = O.vit_block

    def block(t, sd_, p, heads=12):
        B, N, C = t.shape
        ra = "a" in plan.get(p + "attn", ())
        rq = R16 if ra else (lambda z: z)
        h = F.layer_norm(t, (C,), sd_[p + "norm1.weight"], sd_[p + "norm1.bias"], 1e-6)
        qkv = rq(lin(h, sd_[p + "attn.qkv.weight"], sd_[p + "attn.qkv.bias"]))
        qkv = qkv.reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = ((q @ k.transpose(-2, -1)) * ((C // heads) ** -0.5)).softmax(dim=-1)
        h = (rq(attn) @ v).transpose(1, 2).reshape(B, N, C)
        macs[p + "attn"] = float(2 * heads * N * N * (C // heads))
        t = t + lin(h, sd_[p + "attn.proj.weight"], sd_[p + "attn.proj.bias"])
        h = F.layer_norm(t, (C,), sd_[p + "norm2.weight"], sd_[p + "norm2.bias"], 1e-6)
        h = F.gelu(lin(h, sd_[p + "mlp.fc1.weight"], sd_[p + "mlp.fc1.bias"]))
        return t + lin(h, sd_[p + "mlp.fc2.weight"], sd_[p + "mlp.fc2.bias"])

    O.F = Fp()
    O.vit_block = block
    O.std_conv_same = std_conv
    try:
        return O.dpt_forward(sd, x), macs
    finally:
        O.F = F
        O.vit_block = old_block
        O.std_conv_same = std0


def measure(sd, x, verbose=True):
    ref, macs = run(sd, x, {})
    out = {}
    for i, k in enumerate(macs):
        ops = ("a",) if k.endswith(".attn") else ("a", "w")
        rec = {"macs": macs[k]}
        for o in ops:
            y, _ = run(sd, x, {k: {o}})
            d = y - ref
            rec[o] = float(d.pow(2).mean().sqrt())
            rec[o + "_max"] = float(d.abs().max())
        out[k] = rec
        if verbose:
            print(f"[{i + 1}/{len(macs)}] {k:75s} GMAC {macs[k] / 1e9:7.3f}  a {rec['a']:.2e}  w {rec.get('w', 0):.2e}", flush=True)
    return out


def solve(tables, budget_rms, force_x3=(), options=("w2", "a2", "x3")):
    """tables: list of per-family sensitivity dicts (the constraint must hold for each).  Returns {layer: option}.
    Greedy: start at 1 MFMA everywhere; repeatedly apply the upgrade with the largest (worst-family variance
    removed) / (extra MACs) until every family's predicted rms is <= budget."""
    keys = list(tables[0].keys())
    opt = {k: "1" for k in keys}
    for k in keys:
        if any(f in k for f in force_x3):
            opt[k] = "x3"

    def left(t, k, o):  # variance left by option o
        a2, w2 = t[k]["a"] ** 2, t[k].get("w", 0.0) ** 2
        return {"1": a2 + w2, "w2": a2, "a2": w2, "x3": 0.0}[o]

    cost = {"1": 1, "w2": 2, "a2": 2, "x3": 3}

    def total(t):
        return sum(left(t, k, opt[k]) for k in keys)

    while max(total(t) for t in tables) > budget_rms ** 2:
        worst = max(tables, key=total)
        best = None
        for k in keys:
            if k.endswith(".attn"):
                cands = ["x3"] if opt[k] == "1" else []
            else:
                cands = [o for o in {"1": ["w2", "a2", "x3"], "w2": ["x3"], "a2": ["x3"], "x3": []}[opt[k]] if o in options]
            for o in cands:
                gain = left(worst, k, opt[k]) - left(worst, k, o)
                extra = (cost[o] - cost[opt[k]]) * worst[k]["macs"]
                if gain <= 0:
                    continue
                score = gain / extra
                if best is None or score > best[0]:
                    best = (score, k, o)
        if best is None:
            break
        opt[best[1]] = best[2]
    return opt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--family", nargs="+", default=["default", "trained"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--task", default="normal")
    ap.add_argument("--out", default=None)
    ap.add_argument("--load", default=None)
    ap.add_argument("--solve", type=float, nargs="*", default=None)
    ap.add_argument("--force-x3", nargs="*", default=[])
    ap.add_argument("--options", nargs="*", default=["w2", "a2", "x3"], help="upgrades the solver may use")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    C = 1 if args.task == "depth" else 3
    data = json.load(open(args.load)) if args.load else {}
    inputs = {}
    for fam in args.family:
        sd = random_state_dict(args.seed, C, family=fam)
        x = synthetic_input(args.seed, 1, args.task)
        inputs[fam] = (sd, x)
        if fam not in data:
            print(f"== measuring family={fam}", flush=True)
            data[fam] = measure(sd, x)
            if args.out:
                json.dump(data, open(args.out, "w"), indent=1)
    if args.solve:
        tables = [data[f] for f in args.family]
        tot_macs = sum(v["macs"] for v in tables[0].values())
        for budget in args.solve:
            opt = solve(tables, budget, args.force_x3, tuple(args.options))
            cost = {"1": 1, "w2": 2, "a2": 2, "x3": 3}
            rel = sum(cost[opt[k]] * tables[0][k]["macs"] for k in opt) / tot_macs
            hist = {}
            for k, o in opt.items():
                hist[o] = hist.get(o, 0) + 1
            print(f"== budget rms {budget:.2e}: MFMA cost {rel:.3f}x single-pass, options {hist}")
            for k, o in opt.items():
                if o != "1":
                    print(f"     {o:3s} {k}")
            for fam in args.family:
                sd, x = inputs[fam]
                ref, _ = run(sd, x, {})
                plan = {k: {"1": {"a", "w"}, "w2": {"a"}, "a2": {"w"}, "x3": set()}[o] for k, o in opt.items()}
                y, _ = run(sd, x, plan)
                d = (y - ref).abs()
                print(f"   verify family={fam}: max|d| {d.max():.3e} rms {d.pow(2).mean().sqrt():.3e}")


if __name__ == "__main__":
    main()
