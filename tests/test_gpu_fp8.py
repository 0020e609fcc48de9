"""fp8 dtype (BASELINE.json configs[4] "fp8 MFMA weights"): the decoder's RCU / out_conv / first head convolutions on OCP
e4m3 operands.  e4m3 has a 4-bit significand: every operand carries ~2^-4 relative rounding noise, so this is a throughput
mode with ITS OWN stated tolerance, not a parity mode:

  * op level (this file): the fp8 convolution against an fp32 convolution of the SAME e4m3 operands -- the kernel's own
    arithmetic is exact up to fp32 accumulation, so the bf16-output tolerance of the 16-bit kernels applies (6e-3 of
    max|ref|); its e4m3 output copy within one e4m3 ulp.
  * end to end: against the fp32 CPU oracle on the seeded weights, mean angular error < 20 deg / rms < 6e-2 for the
    normal head (CPU emulation of exactly this policy, oracle/precision_policy.py: 13.7 deg / 3.9e-2; bf16 alone is
    3.6 deg / 1.1e-2), and the decoder stage taps grow by a few % rms per fp8 conv, not more.
pytest -m gpu."""
import pytest
import torch
import torch.nn.functional as F

from omnidata_amd.engine import load_library
from omnidata_amd.model import DPTDepthModel, DPTDualTaskModel
from omnidata_amd.weights import random_dual_state_dict, synthetic_input
from tests.gpu_util import OUT_TOL, ptr, rel_err, stream
from tests.test_gpu_e2e import oracle_case
from tests.test_gpu_ops import conv_ref
from oracle.dpt_oracle import dpt_forward_dual, mean_angular_error_deg, oracle_threads

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F8 = torch.float8_e4m3fn


def r8(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(F8).to(DEV)


# B, H, Cin, Cout, k, stride, pad, Ho, act, residual, q_relu
@pytest.mark.parametrize("case", [(2, 24, 256, 256, 3, 1, 1, 24, 1, False, 0),    # RCU conv1 (ReLU epilogue)
                                  (2, 24, 256, 256, 3, 1, 1, 24, 0, True, 1),     # RCU conv2 (+ residual), ReLU'd e4m3 copy
                                  (3, 48, 256, 256, 1, 1, 0, 48, 0, False, 0),    # out_conv 1x1
                                  (2, 48, 256, 128, 3, 1, 1, 48, 0, False, 0),    # first head conv (N = 128)
                                  (8, 96, 256, 256, 3, 1, 1, 96, 0, True, 0),     # big M: 256x256 8-wave tile
                                  (1, 12, 512, 64, 3, 1, 1, 12, 0, False, 1)])    # small map, N = 64, K = 4608
def test_fp8_conv(case):
    lib = load_library()
    B, H, Cin, Cout, k, stride, pad, Ho, act, res, q_relu = case
    X8 = r8(B, H, H, Cin, seed=30)
    W8 = r8(Cout, k, k, Cin, scale=16.0, seed=31)          # weights as the engine stores them: scaled up to the e4m3 range
    out_scale = (1.0 / 16.0) * (k * k * Cin) ** -0.5
    bias = torch.randn(Cout, device=DEV) * 0.1
    R = (torch.randn(B, Ho, Ho, Cout, generator=torch.Generator().manual_seed(32))).to(torch.bfloat16).to(DEV) if res else None
    Y = torch.empty(B, Ho, Ho, Cout, device=DEV, dtype=torch.bfloat16)
    Y8 = torch.zeros(B, Ho, Ho, Cout, device=DEV, dtype=torch.uint8)
    rc = lib.dptx_op_conv_fp8(ptr(X8), ptr(W8), ptr(bias), ptr(R), ptr(Y), ptr(Y8), B, H, H, Cin, Cout, k, stride, pad, pad, Ho, Ho,
                              act, q_relu, out_scale, stream())
    assert rc == 0
    ref = conv_ref(X8.float(), W8.float(), None, stride, pad, pad, Ho, Ho, 0) * out_scale + bias
    if act == 1:
        ref = F.relu(ref)
    if res:
        ref = ref + R.float()
    assert rel_err(Y.float(), ref) < OUT_TOL["bf16"]
    want8 = (F.relu(ref) if q_relu else ref).clamp(-448, 448)
    got8 = Y8.view(F8).float()
    # the copy is the e4m3 rounding of the kernel's fp32 value: it can differ from the rounding of `ref` by one e4m3 step
    # where the two fp32 values straddle a rounding boundary
    step = torch.exp2(torch.floor(torch.log2(want8.abs().clamp_min(2.0 ** -6))) - 3)
    assert ((got8 - want8).abs() <= step * 1.001 + 1e-6).all()
    assert float((got8 == want8.to(F8).float()).float().mean()) > 0.98


def test_fp8_end_to_end_stated_tolerance():
    """The LOSSY preset (fp8_all: all 19 eligible decoder convolutions on e4m3) against the oracle, stage by stage."""
    sd, x, ref, otaps = oracle_case("normal", 3, 0, 1)
    model = DPTDepthModel(num_channels=3, dtype="fp8", max_batch=1, fp8_all=True)
    model.load_state_dict(sd)
    model.to(DEV)
    eng = model._get_engine(torch.device(DEV))
    eng.enable_taps(True)
    y = model(x.to(DEV)).cpu()
    d = (y - ref).abs()
    ang = mean_angular_error_deg(y.clamp(0, 1), ref.clamp(0, 1))
    print(f"\n[fp8_all decoder, bf16 elsewhere] max|d|={d.max():.3e} rms={d.pow(2).mean().sqrt():.3e} mean angular error {ang:.2f} deg")
    rel = {}
    for n in ["s2", "blk11", "l1_rn", "l4_rn", "p4", "p3", "p2", "p1", "h0"]:
        got, want = eng.tap(n), otaps[n]
        rel[n] = ((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
        print(f"    tap {n:6s} rms-rel err {rel[n]:.3e}")
    assert torch.isfinite(y).all() and (y >= 0).all()
    assert ang < 20.0 and d.pow(2).mean().sqrt() < 6e-2
    assert rel["l1_rn"] < 0.2 and max(rel[n] for n in ("p4", "p3", "p2", "p1", "h0")) < 0.35


@pytest.mark.parametrize("family", ["default", "trained"])
def test_fp8_error_on_both_weight_families(family):
    """BASELINE configs[4] 'fp8 MFMA weights', with a bar: the DEFAULT fp8 mode -- the six resConfUnit1
    convolutions of refinenet1..3 on e4m3 operands, chosen layer by layer on the CPU oracle (oracle/fp8_layers.py) -- stays
    within 2 x the bf16 engine's mean angular error on BOTH synthetic weight families (omnidata_amd.weights: 'default' =
    chaotic random residual net, 'trained' = trained-like conditioning).  The fp8 MFMA takes e4m3 on both operands, so
    'fp8 weights' means e4m3 activations as well, 2^-4 relative rounding per element: with all 19 eligible convolutions
    (fp8_all, round 3's mode) the error is 2 - 7 x the bf16 one -- a lossy throughput mode with the loose bar it had."""
    from omnidata_amd.weights import random_state_dict
    from oracle.dpt_oracle import dpt_forward
    sd = random_state_dict(0, 3, family=family)
    x = synthetic_input(0, 1, "normal")
    oracle_threads()
    ref = dpt_forward(sd, x)
    res = {}
    for name, kw in (("bf16", dict(dtype="bf16")), ("fp8", dict(dtype="fp8")), ("fp8_all", dict(dtype="fp8", fp8_all=True))):
        m = DPTDepthModel(num_channels=3, max_batch=1, **kw)
        m.load_state_dict(sd)
        m.to(DEV)
        m.calibrate(x.to(DEV)) if kw["dtype"] == "fp8" else None
        y = m(x.to(DEV)).cpu()
        res[name] = ((y - ref).pow(2).mean().sqrt().item(), mean_angular_error_deg(y.clamp(0, 1), ref.clamp(0, 1)))
    print(f"\n[{family}] bf16 rms {res['bf16'][0]:.3e} / {res['bf16'][1]:.2f} deg;  fp8 (6 safe convs) rms {res['fp8'][0]:.3e} / "
          f"{res['fp8'][1]:.2f} deg;  fp8_all (19 convs) rms {res['fp8_all'][0]:.3e} / {res['fp8_all'][1]:.2f} deg")
    assert res["fp8"][1] <= 2.0 * res["bf16"][1], "the default fp8 mode must stay within 2 x the bf16 mode's angular error"
    assert res["fp8_all"][1] < 20.0 and res["fp8_all"][0] < 6e-2


def test_fp8_activation_scales_follow_the_data():
    """activations were quantised to e4m3 at a fixed unit scale -- fine for the seeded weights, whose
    activations are O(1) by construction, wrong for a checkpoint whose decoder activations leave 2^-9 .. 448.  Here the
    decoder is re-parameterised so that it computes the SAME function with 600x larger internal activations (ReLU is
    positively homogeneous: layerN_rn weights and every refinenet bias x 600, first head conv weight / 600).  With
    calibrated per-tensor scales (dptx_calibrate_fp8; DPTDepthModel does it on its first batch) the result must be as good
    as on the original weights; at unit scale nearly everything would saturate at 448."""
    sd, x, ref, _ = oracle_case("normal", 3, 0, 1)
    big = {k: v.clone() for k, v in sd.items()}
    for k in big:
        if k.startswith("scratch.layer") and k.endswith("_rn.weight"):
            big[k] *= 600.0
        elif k.startswith("scratch.refinenet") and k.endswith(".bias"):
            big[k] *= 600.0
    big["scratch.output_conv.0.weight"] /= 600.0
    outs = {}
    for name, w in (("orig", sd), ("x600", big)):
        m = DPTDepthModel(num_channels=3, dtype="fp8", max_batch=1, fp8_all=True)
        m.load_state_dict(w)
        m.to(DEV)
        outs[name] = m(x.to(DEV)).cpu()
        scales, amax = m.engine.fp8_calibration()
        print(f"\n[{name}] {len(scales)} e4m3 tensors: max|x| {amax.min():.3g} .. {amax.max():.3g}, scales 2^{int(torch.tensor(scales).log2().min())} .. 2^{int(torch.tensor(scales).log2().max())}")
        assert len(scales) >= 19 and (amax > 0).all() and (amax * scales <= 224.0 * 1.0001).all() and (amax * scales > 111.9).all()
    e_orig = (outs["orig"] - ref).pow(2).mean().sqrt().item()
    e_big = (outs["x600"] - ref).pow(2).mean().sqrt().item()
    print(f"    rms vs fp32 oracle: original {e_orig:.3e}, 600x activations {e_big:.3e}")
    assert e_big < 1.5 * e_orig + 5e-3


def test_fp8_dual_task_runs_and_is_deterministic():
    sd = random_dual_state_dict(3)
    x = synthetic_input(11, 3, "normal")
    dual = DPTDualTaskModel(dtype="fp8", max_batch=3, fp8_all=True)
    dual.load_state_dict(sd)
    dual.to(DEV)
    yn, yd = dual(x.to(DEV))
    assert yn.shape == (3, 3, 384, 384) and yd.shape == (3, 384, 384)
    yn2, yd2 = dual(x.to(DEV))
    assert torch.equal(yn, yn2) and torch.equal(yd, yd2)
    oracle_threads()
    rn, rd = dpt_forward_dual(sd, x)
    en, ed = (yn.cpu() - rn), (yd.cpu() - rd)
    print(f"\n[fp8 dual] normal rms {en.pow(2).mean().sqrt():.3e} ang {mean_angular_error_deg(yn.cpu().clamp(0, 1), rn.clamp(0, 1)):.2f} deg; "
          f"depth rms {ed.pow(2).mean().sqrt():.3e}")
    assert en.pow(2).mean().sqrt() < 6e-2 and ed.pow(2).mean().sqrt() < 6e-2


@pytest.mark.parametrize("family", ["default", "trained"])
def test_fp8_vit_error_on_both_weight_families(family):
    """Round 6 (VERDICT r5 item 6 / missing #3: 'fp8 for the encoder'): DPTX_FLAG_FP8_VIT runs qkv / fc1 / fc2 of every
    transformer block -- 45 of the forward's 127.6 GMAC -- on e4m3 operands next to the default six decoder convolutions
    (weights per output channel, ONE calibrated scale per activation tensor, LayerNorm fold on the e4m3 kernel with the column
    sums of the dequantised weights).  oracle/fp8_vit.py emulated these layers on the CPU first: 2.1-2.7 deg (default family) /
    1.2-1.9 deg (trained-like) of mean angular error by themselves, which adds in quadrature to the decoder preset's.  Bar:
    the SAME bar as the default preset's -- the engine with BOTH stays within 2 x the bf16 engine's mean angular error on both
    families (measured: 4.62 vs 4.18 deg, 1.98 vs 1.07 deg) -- and the ViT part alone must not exceed what the emulation
    predicted (<= 3.2 deg / <= 2.3 deg added in quadrature; measured 1.23 / 1.09)."""
    from omnidata_amd.weights import random_state_dict
    from oracle.dpt_oracle import dpt_forward
    sd = random_state_dict(0, 3, family=family)
    x = synthetic_input(0, 1, "normal")
    oracle_threads()
    ref = dpt_forward(sd, x)
    res = {}
    for name, kw in (("bf16", dict(dtype="bf16")), ("fp8", dict(dtype="fp8")), ("fp8+vit", dict(dtype="fp8", fp8_vit=True))):
        m = DPTDepthModel(num_channels=3, max_batch=1, **kw)
        m.load_state_dict(sd)
        m.to(DEV)
        if kw["dtype"] == "fp8":
            m.calibrate(x.to(DEV))
        y = m(x.to(DEV)).cpu()
        assert torch.isfinite(y).all()
        res[name] = ((y - ref).pow(2).mean().sqrt().item(), mean_angular_error_deg(y.clamp(0, 1), ref.clamp(0, 1)))
        if name == "fp8+vit":
            scales, amax = m.engine.fp8_calibration()
            # e4m3 tensors: the token stream after patch-embed, after every proj and every fc2 but the last (the next block's qkv
            # operand), every GELU output, + the decoder preset's six
            assert len(scales) >= 1 + 12 + 11 + 12 + 6 and (amax > 0).all()
    vit_part = max(res["fp8+vit"][1] ** 2 - res["fp8"][1] ** 2, 0.0) ** 0.5
    print(f"\n[{family}] bf16 {res['bf16'][1]:.2f} deg (rms {res['bf16'][0]:.3e});  fp8 (6 decoder convs) {res['fp8'][1]:.2f} deg;  "
          f"fp8 + ViT qkv/fc1/fc2 {res['fp8+vit'][1]:.2f} deg (rms {res['fp8+vit'][0]:.3e}); ViT part in quadrature {vit_part:.2f} deg")
    assert res["fp8+vit"][1] <= 2.0 * res["bf16"][1]
    assert vit_part <= (3.2 if family == "default" else 2.3)


def test_fp8_vit_is_deterministic_batch_invariant_and_schedule_independent():
    """The e4m3 ViT GEMMs keep the engine's invariants: same bits run to run, image i of a batch == the image alone (per-TENSOR
    scales are calibration constants, not data of the batch), two half-batches on two streams == one stream, dual-task == the
    single-task forward of the shared encoder."""
    from omnidata_amd.engine import Engine
    from omnidata_amd.weights import random_state_dict
    sd = random_state_dict(0, 3)
    x = synthetic_input(21, 6, "normal").to(DEV)
    outs = []
    scales = None
    for streams in (1, 2):
        e = Engine(num_channels=3, max_batch=6, dtype="fp8", device_id=0, streams=streams, flags=32)
        e.load_state_dict(sd)
        if scales is None:
            e.calibrate_fp8(x)
            scales = e.fp8_calibration()[0]
        else:
            e.set_fp8_calibration(scales)
        y = e.forward(x).clone()
        assert torch.equal(e.forward(x), y)
        assert torch.equal(e.forward(x[2:3])[0], y[2])            # batch invariance under fixed scales
        outs.append(y)
        e.close()
    assert torch.equal(outs[0], outs[1])
    plain = Engine(num_channels=3, max_batch=6, dtype="fp8", device_id=0, streams=1)   # without the flag: a different result
    plain.load_state_dict(sd)
    plain.calibrate_fp8(x)
    assert not torch.equal(plain.forward(x), outs[0])
    plain.close()
