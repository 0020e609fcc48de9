"""Where does the x3 attention kernel lose precision?  (diagnostic)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omnidata_amd.engine import load_library
from tests.gpu_util import PlaneArena, ptr, rel_err, stream

lib = load_library()
B, S, H = 2, 577, 12
def g(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))
for name, dt, mode in (("bf16x3", torch.bfloat16, 2), ("fp16x3", torch.float16, 3)):
    for case in ("plain", "v_ones", "qk_exact", "v_exact", "all_exact", "small_qk"):
        q0 = g(B, S, 3, H, 64, seed=7)
        if case == "v_ones":
            q0[:, :, 2] = 1.0
        if case in ("qk_exact", "all_exact"):
            q0[:, :, :2] = q0[:, :, :2].to(dt).float()
        if case in ("v_exact", "all_exact"):
            q0[:, :, 2] = q0[:, :, 2].to(dt).float()
        if case == "small_qk":
            q0[:, :, :2] *= 0.25
        ar = PlaneArena(B * S * 4 * H * 64 + 4096, dtype=dt)
        qkv = ar.put(q0.reshape(B * S, 3 * H * 64))
        out = ar.empty(B * S, H * 64)
        assert lib.dptx_op_attention(mode, ptr(qkv), ptr(out), B, S, H, stream()) == 0
        q3 = ar.value(qkv).view(B, S, 3, H, 64)
        q, k, v = [t.permute(0, 2, 1, 3) for t in q3.unbind(2)]
        ref = (((q @ k.transpose(-1, -2)) * 0.125).softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * S, H * 64)
        got = ar.value(out)
        err = (got - ref).abs()
        worst = int(err.argmax())
        print(f"{name} {case:10s} rel err {rel_err(got, ref):.3e}  worst at row {worst // (H * 64)} (token {worst // (H * 64) % S}) col {worst % (H * 64)}"
              f"  rms {float(err.pow(2).mean().sqrt()):.3e}")
        ar.release()
