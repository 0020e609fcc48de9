"""Op-level parity of every HIP kernel against a plain PyTorch fp32 reference of the same op on
the same (already 16-bit-rounded) inputs.  Run on an MI355X: pytest -m gpu."""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from omnidata_amd.engine import DTYPES, load_library
from tests.gpu_util import OUT_TOL, TDT, op_conv, op_gemm, ptr, rel_err, stream

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(*shape, dtype="bf16", scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(TDT[dtype]).to(DEV)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1000, 256, 192), (577 * 3, 768, 768), (18464, 2304, 768),
                                   (4608, 256, 2304), (300, 64, 576), (70000, 64, 64), (5000, 32, 1152), (77, 3072, 768),
                                   (18464, 768, 3072)])  # last: fc2 at B=32 -> 256x256 8-wave tile
def test_gemm_plain(dtype, M, N, K):
    # asymmetric, non-identity operands: a row/col swap or a k-permutation mismatch cannot pass
    A, W = rnd(M, K, dtype=dtype, seed=1), rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=2)
    ref = A.float() @ W.float().t()
    got = op_gemm(dtype, A, W)
    assert rel_err(got.float(), ref) < OUT_TOL[dtype]
    got32 = op_gemm(dtype, A, W, c_fp32=True)
    assert rel_err(got32, ref) < 2e-5


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("M,K", [(1731, 768), (18470, 2048)])  # 2nd: 256x256 8-wave tile (slab epilogue), ragged last m-tile
def test_gemm_epilogues(dtype, M, K):
    N = 768
    A, W = rnd(M, K, dtype=dtype, seed=3), rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=4)
    bias = torch.randn(N, device=DEV)
    R16 = rnd(M, N, dtype=dtype, seed=5)
    R32 = torch.randn(M, N, device=DEV)
    base = A.float() @ W.float().t() + bias
    assert rel_err(op_gemm(dtype, A, W, bias, act=1).float(), F.relu(base)) < OUT_TOL[dtype]
    assert rel_err(op_gemm(dtype, A, W, bias, act=2).float(), F.gelu(base)) < OUT_TOL[dtype]
    assert rel_err(op_gemm(dtype, A, W, bias, R=R16).float(), base + R16.float()) < OUT_TOL[dtype]
    assert rel_err(op_gemm(dtype, A, W, bias, R=R32, c_fp32=True), base + R32) < 2e-5
    # fp32 A operand (ProjectReadout reads the fp32 token stream): rounded to 16-bit while staging
    A32 = torch.randn(M, K, device=DEV)
    ref = A32.to(TDT[dtype]).float() @ W.float().t() + bias
    assert rel_err(op_gemm(dtype, A32, W, bias, c_fp32=True), ref) < 2e-5


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("M,K", [(18464, 768), (18464, 3072), (9232, 768), (1731, 768)])  # proj / fc2 at B = 32, a half batch, a small batch
def test_gemm_token_stream_producer_forms_bitwise(dtype, M, K):
    """proj / fc2 on the 16-bit token stream: C <- C + A W^T + bias IN PLACE plus the LayerNorm fold's row statistics
    (dptx_op_gemm_stream).  Round 6: at B = 32 the launch takes the register-direct epilogue of the 256x256 kernel (residual
    loads up front, the statistics' butterfly replayed across lanes / registers / the two waves of a 128-column block); the
    staged epilogue (debug flag 1) and the one-block-per-tile launch (flag 3) must give the same bits in C AND in the records,
    and both must be right: C against fp32 of the same expression, the records against sums of the fp32 rows."""
    lib = load_library()
    N = 768
    A, W = rnd(M, K, dtype=dtype, seed=11), rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=12)
    bias = torch.randn(N, device=DEV)
    C0 = rnd(M, N, dtype=dtype, seed=13)
    outs = []
    try:
        for flags in (0, 1, 3):
            lib.dptx_debug_set_gemm_flags(flags)
            C = C0.clone()
            stats = torch.full((M, 8, 2), float("nan"), device=DEV)
            rc = lib.dptx_op_gemm_stream(DTYPES[dtype], ptr(A), ptr(W), ptr(bias), ptr(C), ptr(stats), M, N, K, stream())
            assert rc == 0
            outs.append((C, stats))
    finally:
        lib.dptx_debug_set_gemm_flags(0)
    for C, stats in outs[1:]:
        assert torch.equal(C, outs[0][0])
        assert torch.equal(stats[:, :N // 128], outs[0][1][:, :N // 128])
    C, stats = outs[0]
    ref = A.float() @ W.float().t() + bias + C0.float()
    assert rel_err(C.float(), ref) < OUT_TOL[dtype]
    blk = ref.double().view(M, N // 128, 128)
    assert rel_err(stats[:, :N // 128, 0], blk.sum(2)) < 1e-4        # statistics of the fp32 values, not of the rounded stream
    assert rel_err(stats[:, :N // 128, 1], (blk * blk).sum(2)) < 1e-4
    assert torch.isnan(stats[:, N // 128:]).all()                    # records of blocks the launch does not own stay untouched


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("M,K", [(18464, 768), (18464, 3072), (1731, 768)])
def test_gemm_fp32_token_stream_producer_forms_bitwise(dtype, M, K):
    """The parity mode's proj / fc2: fp32 stream in place + its 16-bit image + row statistics (dptx_op_gemm_stream32).  These
    launches stay on the staged epilogue (a register-direct form was built in round 6 and measured slower:
    profiles/r06_experiments.md section 8); the launch forms that exist -- persistent tile loop, one block per tile -- must agree
    bit for bit, and be right."""
    lib = load_library()
    N = 768
    A, W = rnd(M, K, dtype=dtype, seed=21), rnd(N, K, dtype=dtype, scale=K ** -0.5, seed=22)
    bias = torch.randn(N, device=DEV)
    X0 = torch.randn(M, N, device=DEV)
    outs = []
    try:
        for flags in (0, 1, 3):
            lib.dptx_debug_set_gemm_flags(flags)
            X = X0.clone()
            C16 = torch.zeros(M, N, dtype=TDT[dtype], device=DEV)
            stats = torch.full((M, 8, 2), float("nan"), device=DEV)
            rc = lib.dptx_op_gemm_stream32(DTYPES[dtype], ptr(A), ptr(W), ptr(bias), ptr(X), ptr(C16), ptr(stats), M, N, K, stream())
            assert rc == 0
            outs.append((X, C16, stats))
    finally:
        lib.dptx_debug_set_gemm_flags(0)
    for X, C16, stats in outs[1:]:
        assert torch.equal(X, outs[0][0]) and torch.equal(C16, outs[0][1])
        assert torch.equal(stats[:, :N // 128], outs[0][2][:, :N // 128])
    X, C16, stats = outs[0]
    ref = A.float() @ W.float().t() + bias + X0
    assert rel_err(X, ref) < 2e-5
    assert torch.equal(C16, X.to(TDT[dtype]))                        # the 16-bit image is the rounding of the stored fp32 value
    blk = ref.double().view(M, N // 128, 128)
    assert rel_err(stats[:, :N // 128, 0], blk.sum(2)) < 1e-4
    assert rel_err(stats[:, :N // 128, 1], (blk * blk).sum(2)) < 1e-4


def conv_ref(X, Wt, bias, stride, pad_t, pad_l, Ho, Wo, a_relu):
    x = X.float().permute(0, 3, 1, 2)
    if a_relu:
        x = F.relu(x)
    k = Wt.shape[1]
    H, W = x.shape[-2:]
    pb = max((Ho - 1) * stride + k - H - pad_t, 0)
    pr = max((Wo - 1) * stride + k - W - pad_l, 0)
    x = F.pad(x, [pad_l, pr, pad_t, pb])
    y = F.conv2d(x, Wt.float().permute(0, 3, 1, 2), bias, stride)
    assert y.shape[-2:] == (Ho, Wo)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", [
    # B, H, Cin, Cout, k, stride, pad_t, Ho, a_relu, act, residual
    (2, 24, 256, 256, 3, 1, 1, 24, 1, 1, False),   # RCU conv1: pre-ReLU + ReLU epilogue
    (2, 24, 256, 256, 3, 1, 1, 24, 0, 0, True),    # RCU conv2: + residual
    (3, 48, 128, 128, 3, 2, 0, 24, 0, 0, False),   # bottleneck conv2 stride 2, TF-SAME pad (0,1)
    (2, 48, 256, 512, 1, 2, 0, 24, 0, 0, False),   # downsample 1x1 stride 2
    (1, 24, 768, 768, 3, 2, 1, 12, 0, 0, False),   # act_postprocess4 conv 3x3 s2 p1
    (2, 96, 64, 64, 3, 1, 1, 96, 0, 0, False),     # stage0 conv2 (N=64)
    (1, 40, 128, 32, 3, 1, 1, 40, 0, 1, False),    # head conv 128->32 (+ReLU), N=32 tile
    (5, 12, 768, 256, 3, 1, 1, 12, 0, 0, False),   # layer4_rn (small map, K=6912)
    (8, 96, 256, 256, 3, 1, 1, 96, 1, 1, False),   # big-M RCU conv1: 256x256 8-wave tile, pre-ReLU
    (8, 96, 256, 128, 3, 1, 1, 96, 0, 0, True),    # big-M, N=128, residual
    (9, 48, 512, 256, 3, 1, 1, 48, 0, 0, False),   # layer2_rn at batch 9 (M not a multiple of 256)
    (25, 48, 256, 256, 3, 1, 1, 48, 1, 0, True),   # 256x256 tile, M = 57600 (225 m-tiles), pre-ReLU + residual
])
def test_conv_implicit_gemm(dtype, case):
    B, H, Cin, Cout, k, stride, pad, Ho, a_relu, act, res = case
    X = rnd(B, H, H, Cin, dtype=dtype, seed=6)
    Wt = rnd(Cout, k, k, Cin, dtype=dtype, scale=(k * k * Cin) ** -0.5, seed=7)
    bias = torch.randn(Cout, device=DEV) * 0.1
    R = rnd(B, Ho, Ho, Cout, dtype=dtype, seed=8) if res else None
    ref = conv_ref(X, Wt, bias, stride, pad, pad, Ho, Ho, a_relu)
    if act == 1:
        ref = F.relu(ref)
    if res:
        ref = ref + R.float()
    got = op_conv(dtype, X, Wt, bias, R, stride, pad, pad, Ho, Ho, a_relu, act)
    assert rel_err(got.float(), ref) < OUT_TOL[dtype]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,S", [(1, 577), (3, 577), (2, 64), (1, 200), (2, 65), (1, 66), (2, 17), (1, 129)])
def test_attention(dtype, B, S):
    lib = load_library()
    H = 12
    qkv = rnd(B * S, 3 * H * 64, dtype=dtype, seed=9)
    # spike a few keys so the running max really jumps between tiles (online-softmax rescale path)
    q3 = qkv.view(B, S, 3, H, 64)
    q3[:, S // 2, 1] *= 6.0
    q3[:, S - 1, 1] *= 4.0
    q3[0, 0, 1, ::2] *= 5.0   # key 0 (the cls token) is handled outside the key tiles: make it dominate for some heads
    out = torch.empty(B * S, H * 64, device=DEV, dtype=TDT[dtype])
    assert lib.dptx_op_attention(DTYPES[dtype], ptr(qkv), ptr(out), B, S, H, stream()) == 0
    q, k, v = [t.permute(0, 2, 1, 3).float() for t in q3.unbind(2)]
    att = ((q @ k.transpose(-1, -2)) * 0.125).softmax(-1)
    ref = (att @ v).permute(0, 2, 1, 3).reshape(B * S, H * 64)
    # P is rounded to 16 bit before the PV product: allow 2x the output-rounding budget
    assert rel_err(out.float(), ref) < 2 * OUT_TOL[dtype]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_layernorm(dtype):
    lib = load_library()
    M, C = 1155, 768
    x = torch.randn(M, C, device=DEV) * 3 + 0.7
    g, b = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
    y = torch.empty(M, C, device=DEV, dtype=TDT[dtype])
    assert lib.dptx_op_layernorm(DTYPES[dtype], ptr(x), ptr(g), ptr(b), ptr(y), M, C, 1e-6, stream()) == 0
    assert rel_err(y.float(), F.layer_norm(x, (C,), g, b, 1e-6)) < OUT_TOL[dtype]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,HW,C,relu,res", [(2, 9216, 64, 1, False), (2, 2304, 128, 1, False), (3, 576, 1024, 1, True),
                                              (1, 36864, 64, 0, False), (2, 9216, 256, 1, True), (2, 2304, 512, 0, False)])
def test_groupnorm(dtype, B, HW, C, relu, res):
    lib = load_library()
    X = rnd(B, HW, C, dtype=dtype, seed=10) * 2 + 0.5
    X = X.to(TDT[dtype])
    g, b = torch.randn(C, device=DEV), torch.randn(C, device=DEV)
    R = rnd(B, HW, C, dtype=dtype, seed=11) if res else None
    Y = torch.empty_like(X)
    scratch = torch.empty(B * 144 * 64, device=DEV)
    assert lib.dptx_op_groupnorm(DTYPES[dtype], ptr(X), ptr(g), ptr(b), ptr(R), ptr(Y), B, HW, C, relu, 1e-5, ptr(scratch), stream()) == 0
    ref = F.group_norm(X.float().permute(0, 2, 1), 32, g, b, 1e-5).permute(0, 2, 1)
    if res:
        ref = ref + R.float()
    if relu:
        ref = F.relu(ref)
    assert rel_err(Y.float(), ref) < OUT_TOL[dtype]


# (B, H, Cin, Cout, k, stride, pad, Ho, residual): every tile shape the ResNetV2 stage convs take, incl. a ragged last
# m-tile (B*Ho*Ho not a multiple of the tile), images that end inside a tile (576 rows), cpg from 2 to 32
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", [(2, 96, 64, 64, 1, 1, 0, 96, False), (3, 24, 256, 1024, 1, 1, 0, 24, True),
                                  (3, 24, 256, 256, 3, 1, 1, 24, False), (2, 48, 128, 512, 1, 1, 0, 48, True),
                                  (1, 96, 128, 128, 3, 2, 0, 48, False), (5, 24, 1024, 256, 1, 1, 0, 24, False),
                                  (32, 24, 512, 256, 1, 1, 0, 24, False)])
def test_conv_groupnorm_fused_stats(dtype, case):
    """GroupNorm statistics out of the conv's GEMM epilogue (what the ResNetV2 stages run) against conv -> group_norm."""
    lib = load_library()
    B, H, Cin, Cout, k, stride, pad, Ho, res = case
    X = rnd(B, H, H, Cin, dtype=dtype, seed=20)
    Wt = rnd(Cout, k, k, Cin, dtype=dtype, scale=(k * k * Cin) ** -0.5, seed=21)
    g, b = torch.randn(Cout, device=DEV), torch.randn(Cout, device=DEV)
    R = rnd(B, Ho, Ho, Cout, dtype=dtype, seed=22) if res else None
    Yraw = torch.empty(B, Ho, Ho, Cout, device=DEV, dtype=TDT[dtype])
    Y = torch.empty_like(Yraw)
    scratch = torch.zeros(B * (Ho * Ho // 32) * 64, device=DEV)
    rc = lib.dptx_op_conv_groupnorm(DTYPES[dtype], ptr(X), ptr(Wt), ptr(Yraw), ptr(g), ptr(b), ptr(R), ptr(Y), B, H, H, Cin,
                                    Cout, k, stride, pad, pad, Ho, Ho, 1, 1e-5, ptr(scratch), stream())
    assert rc == 0
    raw = conv_ref(X, Wt, None, stride, pad, pad, Ho, Ho, 0)          # fp32 conv of the 16-bit operands, NHWC
    assert rel_err(Yraw.float(), raw) < OUT_TOL[dtype]
    # the statistics are those of the fp32 accumulators; the apply pass normalises the stored (rounded) map
    xr = raw.permute(0, 3, 1, 2).reshape(B, 32, -1)
    mean, var = xr.mean(-1), xr.var(-1, unbiased=False)
    a = (g.view(1, -1) * torch.rsqrt(var + 1e-5).repeat_interleave(Cout // 32, 1))
    ref = Yraw.float() * a.view(B, 1, 1, Cout) + (b.view(1, -1) - mean.repeat_interleave(Cout // 32, 1) * a).view(B, 1, 1, Cout)
    if res:
        ref = ref + R.float()
    ref = F.relu(ref)
    assert rel_err(Y.float(), ref) < OUT_TOL[dtype]
    # batch invariance, bit for bit: image i of the batch == image i alone
    i = B - 1
    Y1 = torch.empty(1, Ho, Ho, Cout, device=DEV, dtype=TDT[dtype])
    R1 = R[i:i + 1].contiguous() if res else None
    rc = lib.dptx_op_conv_groupnorm(DTYPES[dtype], ptr(X[i:i + 1].contiguous()), ptr(Wt), ptr(Y1), ptr(g), ptr(b), ptr(R1), ptr(Y1),
                                    1, H, H, Cin, Cout, k, stride, pad, pad, Ho, Ho, 1, 1e-5, ptr(scratch), stream())
    assert rc == 0
    assert torch.equal(Y1[0], Y[i])


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,H,C", [(2, 12, 256), (1, 96, 256), (2, 48, 128)])
def test_upsample2x_align_corners(dtype, B, H, C):
    lib = load_library()
    X = rnd(B, H, H, C, dtype=dtype, seed=12)
    Y = torch.empty(B, 2 * H, 2 * H, C, device=DEV, dtype=TDT[dtype])
    assert lib.dptx_op_upsample2x(DTYPES[dtype], ptr(X), ptr(Y), B, H, H, C, stream()) == 0
    ref = F.interpolate(X.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True).permute(0, 2, 3, 1)
    assert rel_err(Y.float(), ref) < OUT_TOL[dtype]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,H,W", [(2, 384, 384), (1, 64, 128), (3, 96, 256), (2, 256, 320), (1, 96, 160), (1, 64, 96)])
def test_fused_stem_conv(dtype, B, H, W):
    """7x7 stride-2 TF-SAME conv straight from the NCHW fp32 image (no im2col)."""
    lib = load_library()
    x = torch.rand(B, 3, H, W, device=DEV)
    w = torch.randn(64, 3, 7, 7, device=DEV) * 147 ** -0.5
    wp = torch.zeros(64, 176, device=DEV)
    wp[:, :168].view(64, 3, 7, 8)[..., :7] = w                   # k = (c*7 + ky)*8 + kx
    wp = wp.to(TDT[dtype])
    y = torch.empty(B, H // 2, W // 2, 64, device=DEV, dtype=TDT[dtype])
    assert lib.dptx_op_stem_conv(DTYPES[dtype], ptr(x), ptr(wp), ptr(y), B, H, W, stream()) == 0
    ref = F.conv2d(F.pad(x.to(TDT[dtype]).float(), [2, 3, 2, 3]), w.to(TDT[dtype]).float(), None, 2).permute(0, 2, 3, 1)
    assert ref.shape == y.shape
    assert rel_err(y.float(), ref) < OUT_TOL[dtype]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,Hs,Ws,C,relu", [(2, 48, 64, 3, 1), (1, 192, 192, 1, 1), (3, 32, 48, 3, 0), (5, 16, 16, 2, 1)])
def test_fused_head_tail(dtype, B, Hs, Ws, C, relu):
    """x2 bilinear (align_corners) -> conv3x3 128->32 -> ReLU -> conv1x1 32->C -> ReLU in one kernel (head.hip), against
    the same chain in fp32 torch on the same 16-bit inputs (the up-sampled map rounded to 16 bit, as the kernel does)."""
    lib = load_library()
    H0 = rnd(B, Hs, Ws, 128, dtype=dtype, seed=21)
    W2 = rnd(32, 3, 3, 128, dtype=dtype, scale=1152 ** -0.5, seed=22)
    b2 = torch.randn(32, device=DEV) * 0.3
    w4 = torch.randn(C, 32, device=DEV) * 0.3
    b4 = torch.randn(C, device=DEV) * 0.2
    y = torch.full((B, C, 2 * Hs, 2 * Ws), float("nan"), device=DEV)
    assert lib.dptx_op_head_tail(DTYPES[dtype], ptr(H0), ptr(W2), ptr(b2), ptr(w4), ptr(b4), ptr(y), B, Hs, Ws, C, relu, stream()) == 0
    up = F.interpolate(H0.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=True)
    up = up.to(TDT[dtype]).float()
    h = F.relu(F.conv2d(up, W2.float().permute(0, 3, 1, 2), b2, padding=1))
    ref = F.conv2d(h, w4.view(C, 32, 1, 1), b4)
    if relu:
        ref = F.relu(ref)
    assert torch.isfinite(y).all()
    # a 16-bit rounding of the up-sampled map can flip where the two fp32 interpolation formulas differ in the last bit
    assert rel_err(y, ref) < 2e-3 * (1 if dtype == "bf16" else 0.2)
    # and against this library's own unfused kernels the result is the same up to fp32 summation order
    U = torch.empty(B, 2 * Hs, 2 * Ws, 128, device=DEV, dtype=TDT[dtype])
    assert lib.dptx_op_upsample2x(DTYPES[dtype], ptr(H0), ptr(U), B, Hs, Ws, 128, stream()) == 0
    h2 = F.relu(F.conv2d(U.float().permute(0, 3, 1, 2), W2.float().permute(0, 3, 1, 2), b2, padding=1))
    ref2 = F.conv2d(h2, w4.view(C, 32, 1, 1), b4)
    if relu:
        ref2 = F.relu(ref2)
    assert rel_err(y, ref2) < 2e-5
