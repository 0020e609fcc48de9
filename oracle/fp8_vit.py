"""Do the ViT linears tolerate e4m3 operands?  (TEST INFRASTRUCTURE ONLY -- CPU emulation on the fp32 oracle; round 6.)

VERDICT r5 item 6: "the ViT fc1/fc2 (32.7 GMAC) on e4m3 with per-token scales is where 2x MFMA rate matters; quantify its error on
the oracle first".  Emulated here the way a kernel would do it: weights per OUTPUT CHANNEL after a power-of-two scale into
(224, 448]; activations per TOKEN ROW after a power-of-two scale that puts the row's max |x| into (224, 448] (the producing
epilogue sees whole 128-column blocks of a row, a row maximum is one more reduction there); both rounded to OCP e4m3.  The
normalised activations (what fc1 / qkv consume in the reference) and the un-normalised ones with the LayerNorm folded into the
GEMM (what the engine's kernels consume: (acc - mu colsum) rstd) are both emulated -- the fold subtracts two large numbers.

  python -m oracle.fp8_vit [--out profiles/r06_fp8_vit.json]
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


def q_rows(t):
    """e4m3 with one power-of-two scale per row (last dim = K)."""
    mx = t.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    sc = torch.pow(2.0, torch.floor(torch.log2(448.0 / mx)))
    return (t * sc).clamp(-448.0, 448.0).to(E4M3).float() / sc


def q_tensor(t):
    """e4m3 with ONE power-of-two scale for the whole tensor (what the engine's decoder convolutions use, calibrated)."""
    mx = t.abs().amax().clamp_min(1e-30)
    sc = torch.pow(2.0, torch.floor(torch.log2(448.0 / mx)))
    return (t * sc).clamp(-448.0, 448.0).to(E4M3).float() / sc


def q_blocks(t, blk=32):
    """MX-style: e4m3 with one power-of-two (E8M0) scale per 32 consecutive elements of K -- what
    v_mfma_scale_f32_32x32x64_f8f6f4 applies in hardware; a producer's epilogue thread quad sees exactly such a block."""
    shp = t.shape
    tb = t.reshape(*shp[:-1], shp[-1] // blk, blk)
    mx = tb.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    sc = torch.pow(2.0, torch.floor(torch.log2(448.0 / mx)))
    return ((tb * sc).clamp(-448.0, 448.0).to(E4M3).float() / sc).reshape(shp)


ACT_Q = {"row": None, "tensor": q_tensor, "mx32": q_blocks}


def run(sd, x, kinds, folded, act="row"):
    """fp32 oracle forward with the ViT linears whose key contains one of `kinds` on e4m3 operands.
    folded: quantise the UN-normalised LayerNorm input and fold the LayerNorm into the product, as the engine's GEMMs do."""
    ids = {id(v): k for k, v in sd.items()}
    lin0, ln0 = F.linear, F.layer_norm
    last_ln = {}

    def layer_norm(x_, shape, w, b, eps):
        y = ln0(x_, shape, w, b, eps)
        last_ln["x"], last_ln["w"], last_ln["b"], last_ln["eps"], last_ln["y"] = x_, w, b, eps, y
        return y

    def linear(a, w, b=None):
        k = ids.get(id(w), "")
        if "pretrained.model.blocks." in k and any(s in k for s in kinds):
            if folded and (".qkv." in k or ".fc1." in k) and last_ln.get("y") is a:
                # y = LN(x) W^T + b = ((x W'^T) - mu colsum(W')) rstd + (W beta + b), W' = W * gamma: x and W' on e4m3
                xr, g, be, eps = last_ln["x"], last_ln["w"], last_ln["b"], last_ln["eps"]
                wq = q_rows(w * g[None, :])
                xq = (ACT_Q[act] or q_rows)(xr)
                mu = xr.mean(-1, keepdim=True)
                rstd = torch.rsqrt(xr.var(-1, unbiased=False, keepdim=True) + eps)
                acc = lin0(xq, wq)
                return (acc - mu * wq.sum(1)[None, :]) * rstd + (lin0(be[None, :], w)[0] + b)
            return lin0((ACT_Q[act] or q_rows)(a), q_rows(w), b)
        return lin0(a, w, b)

    class Fp:
        def __getattr__(self, n):
            return {"linear": linear, "layer_norm": layer_norm}.get(n) or getattr(F, n)

    old = O.F
    O.F = Fp()
    try:
        return O.dpt_forward(sd, x)
    finally:
        O.F = old


def ang(y, ref):
    return O.mean_angular_error_deg(y.clamp(0, 1), ref.clamp(0, 1))


def bf16_engine_like(sd, x):
    """Reference point: every ViT linear with bf16-rounded operands (what the bf16 engine does to these layers)."""
    ids = {id(v): k for k, v in sd.items()}
    lin0 = F.linear

    def linear(a, w, b=None):
        if "pretrained.model.blocks." in ids.get(id(w), ""):
            return lin0(a.bfloat16().float(), w.bfloat16().float(), b)
        return lin0(a, w, b)

    class Fp:
        def __getattr__(self, n):
            return linear if n == "linear" else getattr(F, n)
    old = O.F
    O.F = Fp()
    try:
        return O.dpt_forward(sd, x)
    finally:
        O.F = old


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
        yb = bf16_engine_like(sd, x)
        res[fam] = {"bf16 operands in the ViT linears": {"rms": float((yb - ref).pow(2).mean().sqrt()), "ang": float(ang(yb, ref))}}
        print(f"[{fam}] ViT linears on bf16 operands: rms {res[fam]['bf16 operands in the ViT linears']['rms']:.3e} ang {res[fam]['bf16 operands in the ViT linears']['ang']:.3f} deg", flush=True)
        for name, kinds in (("fc2", (".fc2.",)), ("fc1", (".fc1.",)), ("fc1+fc2", (".fc1.", ".fc2.")), ("proj", (".proj.",)),
                            ("qkv", (".qkv.",)), ("all four", (".fc1.", ".fc2.", ".proj.", ".qkv."))):
            for folded in (False, True):
                if folded and not any(k in (".fc1.", ".qkv.") for k in kinds):
                    continue
                y = run(sd, x, kinds, folded)
                key = f"{name} e4m3" + (" (LayerNorm folded: un-normalised x quantised)" if folded else "")
                res[fam][key] = {"rms": float((y - ref).pow(2).mean().sqrt()), "ang": float(ang(y, ref))}
                print(f"[{fam}] {key:70s} rms {res[fam][key]['rms']:.3e}  ang {res[fam][key]['ang']:.3f} deg", flush=True)
        # activation scale granularity (weights per output channel throughout): per tensor (the decoder convolutions' scheme),
        # per token row, per 32-element block of K (MX)
        for act in ("tensor", "mx32"):
            for name, kinds in (("fc1+fc2", (".fc1.", ".fc2.")), ("all four", (".fc1.", ".fc2.", ".proj.", ".qkv."))):
                y = run(sd, x, kinds, True, act)
                key = f"{name} e4m3, activation scale per {act} (LayerNorm folded)"
                res[fam][key] = {"rms": float((y - ref).pow(2).mean().sqrt()), "ang": float(ang(y, ref))}
                print(f"[{fam}] {key:70s} rms {res[fam][key]['rms']:.3e}  ang {res[fam][key]['ang']:.3f} deg", flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
