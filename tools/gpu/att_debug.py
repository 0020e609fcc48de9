"""Where does the fp16x3 attention kernel lose precision?  (diagnostic)  Compares the kernel with fp64 emulations of
its arithmetic under different hypotheses about fp16 subnormal handling."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from omnidata_amd.engine import load_library
from tests.gpu_util import PlaneArena, ptr, rel_err, stream

lib = load_library()
B, S, H = 2, 577, 12
MINN = 2.0 ** -14
def g(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))

def flush(t):
    return torch.where(t.abs() < MINN, torch.zeros_like(t), t)

def emulate(hi, lo, fl_p=False, fl_v=False, fl_qk=False, p_lo=True):
    qh, kh, vh = [t.permute(0, 2, 1, 3).double() for t in hi.unbind(2)]
    ql, kl, vl = [t.permute(0, 2, 1, 3).double() for t in lo.unbind(2)]
    if fl_qk:
        qh, kh, ql, kl = flush(qh), flush(kh), flush(ql), flush(kl)
    if fl_v:
        vh, vl = flush(vh), flush(vl)
    s = (qh @ kh.transpose(-1, -2) + ql @ kh.transpose(-1, -2) + qh @ kl.transpose(-1, -2)).float()
    m = s.max(-1, keepdim=True).values
    p = torch.exp2((s - m) * (0.125 * 1.4426950408889634))
    l = p.sum(-1, keepdim=True).double()
    ph = p.half()
    pl = (p - ph.float()).half()
    ph, pl = ph.double(), pl.double()
    if fl_p:
        ph, pl = flush(ph), flush(pl)
    if not p_lo:
        pl = torch.zeros_like(pl)
    o = (ph @ vh + ph @ vl + pl @ vh) / l
    return o.permute(0, 2, 1, 3).reshape(B * S, H * 64)

q0 = g(B, S, 3, H, 64, seed=7)
ar = PlaneArena(B * S * 4 * H * 64 + 4096, dtype=torch.float16)
qkv = ar.put(q0.reshape(B * S, 3 * H * 64))
out = ar.empty(B * S, H * 64)
assert lib.dptx_op_attention(3, ptr(qkv), ptr(out), B, S, H, stream()) == 0
got = ar.value(out).cpu()
hi = q0.half()
lo = (q0 - hi.float()).half()
val = hi.double() + lo.double()
q, k, v = [t.permute(0, 2, 1, 3) for t in val.unbind(2)]
ref = (((q @ k.transpose(-1, -2)) * 0.125).softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * S, H * 64)
print("kernel vs exact      ", rel_err(got, ref))
for name, kw in (("emu plain", {}), ("emu flush P subnormals", dict(fl_p=True)), ("emu flush V subnormals", dict(fl_v=True)),
                 ("emu flush QK subnormals", dict(fl_qk=True)), ("emu no P_lo", dict(p_lo=False)),
                 ("emu flush P+V+QK", dict(fl_p=True, fl_v=True, fl_qk=True))):
    e = emulate(hi, lo, **kw)
    print(f"{name:28s} vs exact {rel_err(e, ref):.3e}   kernel vs this emulation {rel_err(got, e):.3e}")
err = (got - ref).abs()
top = err.flatten().topk(8)
for val_, idx in zip(top.values, top.indices):
    r, c = int(idx) // (H * 64), int(idx) % (H * 64)
    print(f"  outlier row {r} (img {r // S} token {r % S}) head {c // 64} d {c % 64}: err {float(val_):.3e} got {float(got[r, c]):.6f} ref {float(ref[r, c]):.6f}")
print("elements with err > 1e-6:", int((err > 1e-6).sum()), "of", err.numel())
