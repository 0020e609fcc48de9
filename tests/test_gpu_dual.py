"""Dual-task forward (BASELINE.json configs[4]): normals + depth from ONE encoder pass.
Parity is defined against the reference forward run twice with `pretrained.*` tied (oracle.dpt_forward_dual) and,
bit for bit, against this engine's own single-task forwards on the same tied weights.  pytest -m gpu."""
import pytest
import torch

from omnidata_amd.engine import Engine
from omnidata_amd.model import DPTDepthModel, DPTDualTaskModel
from omnidata_amd.weights import random_dual_state_dict, split_dual_state_dict, synthetic_input
from oracle.dpt_oracle import dpt_forward_dual, mean_angular_error_deg, oracle_threads, ssi_align
from tests.test_gpu_e2e import E2E_TOL

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def single(sd, C, dtype, B):
    m = DPTDepthModel(num_channels=C, dtype=dtype, max_batch=B)
    m.load_state_dict(sd)
    return m.to(DEV)


@pytest.mark.parametrize("dtype", ["bf16", "bf16x3"])
def test_dual_equals_two_single_task_forwards_bitwise(dtype):
    sd = random_dual_state_dict(3)
    nsd, dsd = split_dual_state_dict(sd)
    x = synthetic_input(11, 3, "normal").to(DEV)
    dual = DPTDualTaskModel(dtype=dtype, max_batch=3)
    dual.load_state_dict(sd)
    dual.to(DEV)
    yn, yd = dual(x)
    assert yn.shape == (3, 3, 384, 384) and yd.shape == (3, 384, 384)
    assert torch.equal(yn, single(nsd, 3, dtype, 3)(x))
    assert torch.equal(yd, single(dsd, 1, dtype, 3)(x))
    # MAC accounting: one encoder + two decoders
    n, alg, exe = dual.engine.info()
    assert abs(alg - 185.29e9) < 1e7 and 170e9 < exe < 190e9


@pytest.mark.parametrize("hw", [(384, 384), (256, 320)])
def test_dual_vs_oracle(hw):
    oracle_threads()
    sd = random_dual_state_dict(5)
    x = synthetic_input(12, 1, "normal", hw)
    rn, rd = dpt_forward_dual(sd, x)
    for dtype, tol in (("bf16", E2E_TOL["bf16"][0]), ("bf16x3", 1e-3)):
        dual = DPTDualTaskModel(dtype=dtype, max_batch=1)
        dual.load_state_dict(sd)
        dual.to(DEV)
        yn, yd = [t.cpu() for t in dual(x.to(DEV))]
        assert yn.shape == rn.shape and yd.shape == rd.shape
        dn, dd = (yn - rn).abs().max().item(), (yd - rd).abs().max().item()
        print(f"\n[dual {hw[0]}x{hw[1]} {dtype}] max|d| normal={dn:.3e} depth={dd:.3e}")
        assert dn < tol and dd < tol
        if dtype == "bf16x3":
            assert mean_angular_error_deg(yn.clamp(0, 1), rn.clamp(0, 1)) < 0.05
            assert (ssi_align(yd, rd) - rd).abs().max().item() < 1e-3


@pytest.mark.parametrize("B", [1, 32])
def test_default_dual_model_is_the_parity_mode(B):
    """DPTDualTaskModel defaults to dtype 'mixed' and nothing compared that forward with the oracle.  Both heads
    of the DEFAULT dual model against the reference forward run twice on tied weights (oracle.dpt_forward_dual), at B = 1
    and at B = 32 (two streams, 256x256 tiles) on all 32 images: within north_star's 1e-3."""
    oracle_threads()
    sd = random_dual_state_dict(6)
    x = synthetic_input(13, B, "normal")
    dual = DPTDualTaskModel(max_batch=B)
    assert dual.engine_dtype == "mixed"
    dual.load_state_dict(sd)
    dual.to(DEV)
    yn, yd = [t.cpu() for t in dual(x.to(DEV))]
    assert dual.engine_dtype == "mixed"        # the range guard saw nothing
    dn = dd = 0.0
    for i in range(0, B, 8):                   # the oracle in chunks of 8 images (memory)
        rn, rd = dpt_forward_dual(sd, x[i:i + 8])
        dn = max(dn, (yn[i:i + 8] - rn).abs().max().item())
        dd = max(dd, (yd[i:i + 8] - rd).abs().max().item())
    print(f"\n[dual mixed B={B}] max|d| normal={dn:.3e} depth={dd:.3e}")
    assert dn < 1e-3 and dd < 1e-3


def test_dual_model_leaves_fp16_planes_when_they_overflow():
    """the dual-task model has the fp16 range guard of DPTDepthModel.  Both decoders re-parameterised to
    1e8 x larger internal activations (same function): the default dual model falls back to bf16x3 and matches the oracle."""
    import warnings
    oracle_threads()
    sd = random_dual_state_dict(6)
    x = synthetic_input(13, 1, "normal")
    rn, rd = dpt_forward_dual(sd, x)
    big = {k: v.clone() for k, v in sd.items()}
    for k in big:
        kk = k[6:] if k.startswith("depth.") else k
        if kk.startswith("scratch.layer") and kk.endswith("_rn.weight"):
            big[k] *= 1.0e8
        if kk.startswith("scratch.refinenet") and kk.endswith(".bias"):
            big[k] *= 1.0e8
        if kk == "scratch.output_conv.0.weight":
            big[k] /= 1.0e8
    dual = DPTDualTaskModel(max_batch=1)
    dual.load_state_dict(big)
    dual.to(DEV)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        yn, yd = [t.cpu() for t in dual(x.to(DEV))]
    assert dual.engine_dtype == "bf16x3" and any("fp16 range" in str(i.message) for i in w)
    dn, dd = (yn - rn).abs().max().item(), (yd - rd).abs().max().item()
    print(f"\n[dual 1e8x activations -> bf16x3] max|d| normal={dn:.3e} depth={dd:.3e}")
    assert dn < 1e-3 and dd < 1e-3


def test_dual_taps_and_contract():
    sd = random_dual_state_dict(3)
    dual = DPTDualTaskModel(dtype="bf16", max_batch=2)
    dual.load_state_dict(sd)
    dual.to(DEV)
    x = synthetic_input(11, 2, "normal").to(DEV)
    eng = dual._get_engine(torch.device(DEV))
    eng.enable_taps(True)
    dual(x)
    assert eng.tap("s0").shape == (2, 256, 96, 96) and eng.tap("depth.p1").shape == (2, 256, 192, 192)
    with pytest.raises(RuntimeError, match="unknown or unavailable tap"):
        eng.tap("p1")  # the depth decoder re-used the normal decoder's buffers
    with pytest.raises(RuntimeError, match="dptx_forward_dual"):
        eng.forward(x)
    one = Engine(num_channels=3, max_batch=1, dtype="bf16", device_id=0)
    with pytest.raises(RuntimeError, match="dual_task"):
        one.forward_dual(x[:1])
