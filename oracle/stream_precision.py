"""What does a 16-bit residual stream cost?  (TEST INFRASTRUCTURE ONLY)

The single-pass dtypes (bf16 / fp16 / fp8) keep the token stream of the 12 ViT blocks only as the 16-bit tensor that the
qkv / fc1 GEMMs multiply (engine.hip `stream16`; include/dptx.h DPTX_FLAG_FP32_STREAM switches the fp32 copy back on).  This
script emulates that on the fp32 oracle: every conv / linear / attention operand is rounded to the dtype (as
oracle/precision_policy.py does for "bf16 / fp16 everywhere") and, optionally, the stream is rounded after each of its 24
updates.  Result (seed 0, one image; max abs d / rms against the fp32 forward):

    default bf16: fp32 stream 5.51e-02 / 1.090e-02 | 16-bit stream 5.53e-02 / 1.102e-02
    default fp16: fp32 stream 7.21e-03 / 1.516e-03 | 16-bit stream 6.86e-03 / 1.519e-03
    trained bf16: fp32 stream 1.39e-02 / 2.311e-03 | 16-bit stream 1.47e-02 / 2.440e-03
    trained fp16: fp32 stream 1.62e-03 / 3.326e-04 | 16-bit stream 1.87e-03 / 3.300e-04

i.e. +1 ... 5 % of a deviation that is 9 ... 60x over north_star's 1e-3 either way: a throughput-mode decision.

Usage: python -m oracle.stream_precision
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle.dpt_oracle as O  # noqa: E402
from omnidata_amd.weights import random_state_dict, synthetic_input  # noqa: E402

TYPES = {"bf16": torch.bfloat16, "fp16": torch.float16}


def run(dt, stream_round, fam, seed=0, nimg=1):
    sd = random_state_dict(0, 3, family=fam)
    x = synthetic_input(seed, nimg, "normal")
    ref = O.dpt_forward(sd, x)
    tdt = TYPES[dt]
    r16 = lambda t: t.to(tdt).float()  # noqa: E731
    conv0, lin0 = F.conv2d, F.linear

    def conv(a, w, b=None, *aa, **k):
        return conv0(r16(a), r16(w), b, *aa, **k)

    def lin(a, w, b=None):
        return lin0(r16(a), r16(w), b)

    class Fp:
        def __getattr__(self, n):
            return {"conv2d": conv, "linear": lin}.get(n, getattr(F, n))

    def block(t, sd_, p, heads=12):
        B, N, C = t.shape
        sr = r16 if stream_round else (lambda z: z)
        h = F.layer_norm(t, (C,), sd_[p + "norm1.weight"], sd_[p + "norm1.bias"], 1e-6)
        qkv = r16(lin(h, sd_[p + "attn.qkv.weight"], sd_[p + "attn.qkv.bias"]))
        qkv = qkv.reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = ((q @ k.transpose(-2, -1)) * 0.125).softmax(dim=-1)
        h = (r16(attn) @ v).transpose(1, 2).reshape(B, N, C)
        t = sr(t + lin(h, sd_[p + "attn.proj.weight"], sd_[p + "attn.proj.bias"]))
        h = F.layer_norm(t, (C,), sd_[p + "norm2.weight"], sd_[p + "norm2.bias"], 1e-6)
        h = F.gelu(lin(h, sd_[p + "mlp.fc1.weight"], sd_[p + "mlp.fc1.bias"]))
        return sr(t + lin(h, sd_[p + "mlp.fc2.weight"], sd_[p + "mlp.fc2.bias"]))

    O.F = Fp()
    old = O.vit_block
    O.vit_block = block
    try:
        y = O.dpt_forward(sd, x)
    finally:
        O.F = F
        O.vit_block = old
    d = (y - ref).abs()
    return d.max().item(), d.pow(2).mean().sqrt().item()


def main():
    torch.set_num_threads(os.cpu_count())
    for fam in ("default", "trained"):
        for dt in ("bf16", "fp16"):
            a, b = run(dt, False, fam), run(dt, True, fam)
            print(f"{fam} {dt}: fp32 stream {a[0]:.2e} / {a[1]:.3e} | 16-bit stream {b[0]:.2e} / {b[1]:.3e}")


if __name__ == "__main__":
    main()
