"""What does 16-bit MFMA-operand rounding cost end to end?  (TEST INFRASTRUCTURE ONLY)

Emulates on CPU the engine's numerics: every conv / linear / attention matmul rounds its
two operands to bf16 or fp16 and accumulates in fp32; everything else stays fp32.  Prints
max-abs / rms deviation of the final output against the pure fp32 oracle.
Usage: python -m oracle.precision_study
"""
import os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle.dpt_oracle as O
from omnidata_amd.weights import random_state_dict, synthetic_input


def run(dtype, seed=0, task="normal", C=3, act_round=False):
    r = (lambda t: t.to(dtype).float()) if dtype is not None else (lambda t: t)
    conv0, lin0 = F.conv2d, F.linear
    def conv(x, w, b=None, *a, **k):
        return conv0(r(x), r(w), b, *a, **k)
    def lin(x, w, b=None):
        return lin0(r(x), r(w), b)
    class Fp:  # proxy for F inside the oracle module
        def __getattr__(self, n):
            return {"conv2d": conv, "linear": lin}.get(n, getattr(F, n))
    O.F = Fp()
    old_block = O.vit_block
    def block(x, sd, p):
        B, N, Cc = x.shape
        h = F.layer_norm(x, (Cc,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
        qkv = r(lin(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]))
        qkv = qkv.reshape(B, N, 3, 12, 64).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = ((q @ k.transpose(-2, -1)) * 0.125).softmax(dim=-1)
        h = (r(attn) @ v).transpose(1, 2).reshape(B, N, Cc)
        x = x + lin(h, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        h = F.layer_norm(x, (Cc,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
        h = F.gelu(lin(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        return x + lin(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    O.vit_block = block
    try:
        sd = random_state_dict(seed, C)
        x = synthetic_input(seed, 1, task)
        return O.dpt_forward(sd, x)
    finally:
        O.F = F
        O.vit_block = old_block


if __name__ == "__main__":
    torch.set_num_threads(os.cpu_count())
    for task, C in (("normal", 3), ("depth", 1)):
        ref = run(None, task=task, C=C)
        for dt in (torch.bfloat16, torch.float16):
            y = run(dt, task=task, C=C)
            d = (y - ref).abs()
            print(f"{task:6s} {str(dt):15s} max|d|={d.max():.3e} rms={d.pow(2).mean().sqrt():.3e} "
                  f"p99.9={d.flatten().kthvalue(int(0.999 * d.numel())).values:.3e} (out std {ref.std():.3f})")


def stage_errors(dtype, names=("stem", "s0", "s1", "s2", "tok0", "blk11", "l3", "l4", "p4", "p1", "h0", "h1")):
    """rms-relative error of the emulated 16-bit-operand forward at every stage tap."""
    import oracle.dpt_oracle as OO
    sd = random_state_dict(0, 3)
    x = synthetic_input(0, 1, "normal")
    ref = {}
    OO.dpt_forward(sd, x, ref)
    got = {}
    r = lambda t: t.to(dtype).float()
    conv0, lin0 = F.conv2d, F.linear
    class Fp:
        def __getattr__(self, n):
            return {"conv2d": lambda a, w, b=None, *aa, **k: conv0(r(a), r(w), b, *aa, **k),
                    "linear": lambda a, w, b=None: lin0(r(a), r(w), b)}.get(n, getattr(F, n))
    OO.F = Fp()
    try:
        OO.dpt_forward(sd, x, got)
    finally:
        OO.F = F
    return {n: float((got[n] - ref[n]).pow(2).mean().sqrt() / ref[n].pow(2).mean().sqrt()) for n in names}
