"""bf16x3 (hi/lo plane, 3-MFMA) precision mode: op-level checks of the split arithmetic and the
end-to-end 1e-3 gate of north_star against the fp32 CPU oracle.  pytest -m gpu."""
import pytest
import torch
import torch.nn.functional as F

from omnidata_amd.engine import load_library
from omnidata_amd.model import DPTDepthModel
from tests.gpu_util import PlaneArena, ptr, rel_err, stream
from tests.test_gpu_e2e import oracle_case
from oracle.dpt_oracle import mean_angular_error_deg, ssi_align

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
X3 = 2
TOL = 6e-5  # 2^-17 operand split + dropped lo*lo term + 2^-17 output split, relative to max|ref|


def g(*shape, scale=1.0, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("M,N,K", [(1000, 256, 192), (1731, 768, 768), (5000, 32, 1152), (300, 64, 576)])
def test_x3_gemm(M, N, K):
    lib = load_library()
    ar = PlaneArena(M * K + N * K + 2 * M * N + 4096)
    try:
        A, W, R = ar.put(g(M, K, seed=1)), ar.put(g(N, K, scale=K ** -0.5, seed=2)), ar.put(g(M, N, seed=3))
        C = ar.empty(M, N)
        bias = torch.randn(N, device=DEV)
        assert lib.dptx_op_gemm(X3, ptr(A), ptr(W), ptr(bias), ptr(R), ptr(C), M, N, K, 2, 0, 0, 0, stream()) == 0
        ref = F.gelu(ar.value(A) @ ar.value(W).t() + bias.double()) + ar.value(R)
        assert rel_err(ar.value(C), ref) < TOL
        C32 = torch.empty(M, N, device=DEV)
        assert lib.dptx_op_gemm(X3, ptr(A), ptr(W), None, None, ptr(C32), M, N, K, 0, 0, 1, 0, stream()) == 0
        assert rel_err(C32, ar.value(A) @ ar.value(W).t()) < TOL
    finally:
        ar.release()


@pytest.mark.parametrize("case", [(2, 24, 256, 256, 3, 1, 1, 24, 1, 1), (3, 48, 128, 128, 3, 2, 0, 24, 0, 0),
                                  (2, 48, 256, 512, 1, 2, 0, 24, 0, 0), (1, 40, 128, 32, 3, 1, 1, 40, 0, 1)])
def test_x3_conv(case):
    from tests.test_gpu_ops import conv_ref
    lib = load_library()
    B, H, Cin, Cout, k, stride, pad, Ho, a_relu, act = case
    ar = PlaneArena(B * H * H * Cin + Cout * k * k * Cin + 2 * B * Ho * Ho * Cout + 4096)
    try:
        X, Wt = ar.put(g(B, H, H, Cin, seed=4)), ar.put(g(Cout, k, k, Cin, scale=(k * k * Cin) ** -0.5, seed=5))
        R, Y = ar.put(g(B, Ho, Ho, Cout, seed=6)), ar.empty(B, Ho, Ho, Cout)
        bias = torch.randn(Cout, device=DEV) * 0.1
        assert lib.dptx_op_conv(X3, ptr(X), ptr(Wt), ptr(bias), ptr(R), ptr(Y), B, H, H, Cin, Cout, k, stride, pad, pad, Ho, Ho,
                                a_relu, act, stream()) == 0
        ref = conv_ref(ar.value(X), ar.value(Wt), bias, stride, pad, pad, Ho, Ho, a_relu).double()
        if act == 1:
            ref = F.relu(ref)
        ref = ref + ar.value(R)
        assert rel_err(ar.value(Y), ref) < TOL
    finally:
        ar.release()


@pytest.mark.parametrize("B,S", [(2, 577), (1, 200)])
def test_x3_attention(B, S):
    lib = load_library()
    H = 12
    ar = PlaneArena(B * S * 3 * H * 64 + B * S * H * 64 + 4096)
    try:
        q0 = g(B, S, 3, H, 64, seed=7)
        q0[:, S // 2, 1] *= 6.0
        qkv = ar.put(q0.reshape(B * S, 3 * H * 64))
        out = ar.empty(B * S, H * 64)
        assert lib.dptx_op_attention(X3, ptr(qkv), ptr(out), B, S, H, stream()) == 0
        q3 = ar.value(qkv).view(B, S, 3, H, 64)
        q, k, v = [t.permute(0, 2, 1, 3) for t in q3.unbind(2)]
        ref = (((q @ k.transpose(-1, -2)) * 0.125).softmax(-1) @ v).permute(0, 2, 1, 3).reshape(B * S, H * 64)
        assert rel_err(ar.value(out), ref) < 2 * TOL
    finally:
        ar.release()


def test_x3_norms_and_upsample():
    lib = load_library()
    B, HW, C, M = 2, 2304, 128, 600
    ar = PlaneArena(3 * B * HW * C + M * 768 + 5 * B * 24 * 24 * 256 + 4096)
    try:
        X, R, Y = ar.put(g(B, HW, C, seed=8) * 2 + 0.5), ar.put(g(B, HW, C, seed=9)), ar.empty(B, HW, C)
        gm, bt = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
        scratch = torch.empty(B * 144 * 64, device=DEV)
        assert lib.dptx_op_groupnorm(X3, ptr(X), ptr(gm), ptr(bt), ptr(R), ptr(Y), B, HW, C, 1, 1e-5, ptr(scratch), stream()) == 0
        ref = F.relu(F.group_norm(ar.value(X).permute(0, 2, 1), 32, gm.double(), bt.double(), 1e-5).permute(0, 2, 1) + ar.value(R))
        assert rel_err(ar.value(Y), ref) < TOL
        x = torch.randn(M, 768, device=DEV) * 3 + 0.7
        g2, b2 = torch.randn(768, device=DEV), torch.randn(768, device=DEV)
        y = ar.empty(M, 768)
        assert lib.dptx_op_layernorm(X3, ptr(x), ptr(g2), ptr(b2), ptr(y), M, 768, 1e-6, stream()) == 0
        assert rel_err(ar.value(y), F.layer_norm(x.double(), (768,), g2.double(), b2.double(), 1e-6)) < TOL
        U, V = ar.put(g(B, 24, 24, 256, seed=10)), ar.empty(B, 48, 48, 256)
        assert lib.dptx_op_upsample2x(X3, ptr(U), ptr(V), B, 24, 24, 256, stream()) == 0
        ref = F.interpolate(ar.value(U).permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
        assert rel_err(ar.value(V), ref) < TOL
    finally:
        ar.release()


@pytest.mark.parametrize("task,C,seed,B", [("normal", 3, 0, 1), ("depth", 1, 0, 1), ("normal", 3, 1, 2)])
def test_x3_end_to_end_meets_1e3(task, C, seed, B):
    """north_star: outputs within 1e-3 abs of the PyTorch-CPU fp32 forward (normal map / depth)."""
    sd, x, ref, _ = oracle_case(task, C, seed, B)
    model = DPTDepthModel(num_channels=C, dtype="bf16x3", max_batch=B)
    model.load_state_dict(sd)
    model.to(DEV)
    eng = model._get_engine(torch.device(DEV))
    eng.enable_taps(True)
    y = model(x.to(DEV)).cpu()
    d = (y - ref).abs()
    print(f"\n[{task} seed={seed} B={B} bf16x3] max|d|={d.max():.3e} rms={d.pow(2).mean().sqrt():.3e}")
    _, _, _, otaps = oracle_case(task, C, seed, B)
    for n in ["stem", "s0", "s1", "s2", "tok0", "blk0", "blk8", "blk11", "l3", "l4", "l1_rn", "l4_rn", "p4", "p3", "p2", "p1", "h0", "h1"]:
        got, want = eng.tap(n), otaps[n]
        print(f"    tap {n:6s} rms-rel err {((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item():.3e}")
    assert y.shape == ref.shape and torch.isfinite(y).all()
    assert d.max().item() < 1e-3
    if task == "normal":
        assert mean_angular_error_deg(y.clamp(0, 1), ref.clamp(0, 1)) < 0.05
    else:
        assert (ssi_align(y, ref) - ref).abs().max().item() < 1e-3
