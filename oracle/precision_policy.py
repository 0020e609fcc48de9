"""Per-layer-group precision policies, emulated on CPU.  (TEST INFRASTRUCTURE ONLY)

The engine's mixed-precision modes choose, per layer group, how the two operands of every MFMA
are represented: one 16-bit value (bf16 / fp16: relative rounding 2^-9 / 2^-12) or a hi/lo pair of
16-bit planes with three MFMAs per product (~2^-17..2^-22: indistinguishable from fp32 here).  This
script emulates exactly that on the fp32 oracle -- operands of every conv / linear / attention
matmul of a group are rounded to the group's type, accumulation and everything else stay fp32 --
and prints the end-to-end deviation from the pure fp32 oracle for a list of policies, on both
synthetic weight families (omnidata_amd.weights.random_state_dict(family=...)).

It answers, without a GPU: which groups have to run in the 3-MFMA representation for the whole
forward to meet north_star's 1e-3, and what single-pass 16-bit arithmetic costs on
well-conditioned ("trained-like") weights.

Usage: python -m oracle.precision_policy [--family default|trained] [--seeds 0 1] [--task normal]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle.dpt_oracle as O  # noqa: E402
from omnidata_amd.weights import random_state_dict, synthetic_input  # noqa: E402

GROUPS = ("resnet", "embed", "vit", "reassemble", "rn", "fusion", "head")
TYPES = {"fp32": None, "x3": None, "fp16": torch.float16, "bf16": torch.bfloat16}


def group_of_key(key: str) -> str:
    if "patch_embed.backbone" in key:
        return "resnet"
    if "patch_embed.proj" in key:
        return "embed"
    if key.startswith("pretrained.model.blocks"):
        return "vit"
    if key.startswith("pretrained.act_postprocess"):
        return "reassemble"
    if "_rn." in key:
        return "rn"
    if "refinenet" in key:
        return "fusion"
    if "output_conv" in key:
        return "head"
    raise KeyError(key)


def run(policy, sd, x):
    """policy: {group: 'fp32'|'x3'|'fp16'|'bf16'}; missing groups -> fp32."""
    ids = {id(v): k for k, v in sd.items()}
    ctx = {"group": "resnet"}  # standardised StdConv weights are new tensors: only the backbone makes those

    def rounder(w):
        key = ids.get(id(w))
        g = group_of_key(key) if key is not None else ctx["group"]
        dt = TYPES[policy.get(g, "fp32")]
        return (lambda t: t) if dt is None else (lambda t: t.to(dt).float())

    conv0, lin0 = F.conv2d, F.linear

    def conv(a, w, b=None, *aa, **k):
        r = rounder(w)
        return conv0(r(a), r(w), b, *aa, **k)

    def lin(a, w, b=None):
        r = rounder(w)
        return lin0(r(a), r(w), b)

    class Fp:
        def __getattr__(self, n):
            return {"conv2d": conv, "linear": lin}.get(n, getattr(F, n))

    old_block = O.vit_block
    vdt = TYPES[policy.get("vit", "fp32")]
    rv = (lambda t: t) if vdt is None else (lambda t: t.to(vdt).float())

    def block(t, sd_, p):  # attention matmuls round q, k, v and the probabilities like the engine does
        B, N, C = t.shape
        h = F.layer_norm(t, (C,), sd_[p + "norm1.weight"], sd_[p + "norm1.bias"], 1e-6)
        qkv = rv(lin(h, sd_[p + "attn.qkv.weight"], sd_[p + "attn.qkv.bias"]))
        qkv = qkv.reshape(B, N, 3, 12, 64).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = ((q @ k.transpose(-2, -1)) * 0.125).softmax(dim=-1)
        h = (rv(attn) @ v).transpose(1, 2).reshape(B, N, C)
        t = t + lin(h, sd_[p + "attn.proj.weight"], sd_[p + "attn.proj.bias"])
        h = F.layer_norm(t, (C,), sd_[p + "norm2.weight"], sd_[p + "norm2.bias"], 1e-6)
        h = F.gelu(lin(h, sd_[p + "mlp.fc1.weight"], sd_[p + "mlp.fc1.bias"]))
        return t + lin(h, sd_[p + "mlp.fc2.weight"], sd_[p + "mlp.fc2.bias"])

    O.F = Fp()
    O.vit_block = block
    try:
        taps = {}
        y = O.dpt_forward(sd, x, taps)
        return y, taps
    finally:
        O.F = F
        O.vit_block = old_block


POLICIES = {
    "bf16 everywhere": {g: "bf16" for g in GROUPS},
    "fp16 everywhere": {g: "fp16" for g in GROUPS},
    "resnet x3, rest fp16": {**{g: "fp16" for g in GROUPS}, "resnet": "x3"},
    "resnet+embed x3, rest fp16": {**{g: "fp16" for g in GROUPS}, "resnet": "x3", "embed": "x3"},
    "resnet+embed+vit x3, rest fp16": {**{g: "fp16" for g in GROUPS}, "resnet": "x3", "embed": "x3", "vit": "x3"},
    "only resnet fp16": {"resnet": "fp16"},
    "only vit fp16": {"vit": "fp16"},
    "only decoder (rn+fusion+head) fp16": {"rn": "fp16", "fusion": "fp16", "head": "fp16"},
    "only head fp16": {"head": "fp16"},
    "resnet x3, rest bf16": {**{g: "bf16" for g in GROUPS}, "resnet": "x3"},
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--family", nargs="+", default=["default", "trained"])
    ap.add_argument("--seeds", type=int, nargs="+", default=[0])
    ap.add_argument("--task", default="normal")
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    C = 1 if args.task == "depth" else 3
    for fam in args.family:
        for seed in args.seeds:
            sd = random_state_dict(seed, C, family=fam)
            x = synthetic_input(seed, 1, args.task)
            ref, rtaps = run({}, sd, x)
            print(f"== family={fam} seed={seed} task={args.task}: out mean {ref.mean():.3f} std {ref.std():.3f} "
                  f"zeros {float((ref == 0).float().mean()):.3f}")
            for name, pol in POLICIES.items():
                if args.only and not any(o in name for o in args.only):
                    continue
                y, taps = run(pol, sd, x)
                d = (y - ref).abs()
                s2 = float((taps["s2"] - rtaps["s2"]).pow(2).mean().sqrt() / rtaps["s2"].pow(2).mean().sqrt())
                b11 = float((taps["blk11"] - rtaps["blk11"]).pow(2).mean().sqrt() / rtaps["blk11"].pow(2).mean().sqrt())
                extra = ""
                if args.task == "normal":
                    extra = f" ang {O.mean_angular_error_deg(y.clamp(0, 1), ref.clamp(0, 1)):.3f} deg"
                print(f"  {name:36s} max|d|={d.max():.3e} rms={d.pow(2).mean().sqrt():.3e} "
                      f"p99.9={d.flatten().kthvalue(int(0.999 * d.numel())).values:.3e} | s2 {s2:.2e} blk11 {b11:.2e}{extra}")


if __name__ == "__main__":
    main()
