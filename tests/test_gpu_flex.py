"""Inputs that are not 384x384: the reference's forward_flex (vit.py:119-155) resizes
pos_embed to the input's patch grid and everything else is convolutional; dptx_forward_hw does the same on
the GPU.  Parity against the fp32 CPU oracle and against golden vectors written by the reference's own modules
at 256x320 and 448x288 (tests/golden/flex_*.npz).  pytest -m gpu."""
import glob
import os

import numpy as np
import pytest
import torch

from omnidata_amd.model import DPTDepthModel
from omnidata_amd.weights import random_state_dict, synthetic_input
from oracle.dpt_oracle import dpt_forward, oracle_threads
from oracle.validate_vs_reference import subsample
from tests.test_gpu_e2e import E2E_TOL, STAGE_RMS_REL

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
FLEX = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "flex_*.npz")))
_oracle = {}


def oracle_case(task, C, seed, B, hw):
    key = (task, seed, B, hw)
    if key not in _oracle:
        oracle_threads()
        sd = random_state_dict(seed, C)
        x = synthetic_input(seed, B, task, hw)
        taps = {}
        _oracle[key] = (sd, x, dpt_forward(sd, x, taps), taps)
    return _oracle[key]


def make_model(sd, C, dtype, B):
    m = DPTDepthModel(num_channels=C, dtype=dtype, max_batch=B)
    m.load_state_dict(sd)
    return m.to(DEV)


def test_flex_goldens_present():
    assert len(FLEX) >= 2


@pytest.mark.parametrize("path", FLEX, ids=lambda p: os.path.basename(p))
def test_flex_vs_reference_golden(path):
    g = np.load(path)
    task, C, seed, B = str(g["task"]), int(g["num_channels"]), int(g["seed"]), int(g["batch"])
    H, W = int(g["height"]), int(g["width"])
    sd = random_state_dict(seed, C)
    x = synthetic_input(seed, B, task, (H, W)).to(DEV)
    y16 = make_model(sd, C, "fp16", B)(x).cpu()
    assert tuple(y16.shape) == ((B, 3, H, W) if C == 3 else (B, H, W))
    d = np.abs(subsample(y16) - g["out_sub"])
    print(f"\n[{os.path.basename(path)} fp16] max|d|={d.max():.3e}")
    assert d.max() < E2E_TOL["fp16"][0] and np.sqrt((d ** 2).mean()) < E2E_TOL["fp16"][1]
    y3 = make_model(sd, C, "bf16x3", B)(x).cpu()
    d3 = np.abs(subsample(y3) - g["out_sub"])
    row = np.abs(y3.reshape(B, -1, H, W)[0, 0, H // 2 - 1].numpy() - g["out_row"])
    print(f"[{os.path.basename(path)} bf16x3] max|d|={max(d3.max(), row.max()):.3e}")
    assert d3.max() < 1e-3 and row.max() < 1e-3  # north_star tolerance


@pytest.mark.parametrize("task,C,seed,B,hw", [("normal", 3, 6, 2, (192, 512)), ("depth", 1, 7, 1, (512, 384)),
                                              ("normal", 3, 8, 3, (64, 96))])
def test_flex_vs_oracle(task, C, seed, B, hw):
    sd, x, ref, otaps = oracle_case(task, C, seed, B, hw)
    # bf16: the max over ~1e5..1e6 outputs of a heavy-tailed rounding error moves between 4e-2 and 8.5e-2 with seed and
    # size (E2E_TOL's 8e-2 was set on the three 384x384 cases); the rms is the stable figure and keeps E2E_TOL's budget
    for dtype, tol in (("bf16", 1.2e-1), ("bf16x3", 1e-3)):
        model = make_model(sd, C, dtype, B)
        eng = model._get_engine(torch.device(DEV)) if hw[0] * hw[1] <= 384 * 384 else None
        y = model(x.to(DEV)).cpu()
        assert y.shape == ref.shape and torch.isfinite(y).all()
        d = (y - ref).abs().max().item()
        rms = (y - ref).pow(2).mean().sqrt().item()
        print(f"\n[{task} {hw[0]}x{hw[1]} B={B} {dtype}] max|d|={d:.3e} rms={rms:.3e}")
        assert d < tol and rms < E2E_TOL["bf16"][1]
        if eng is not None and dtype == "bf16":  # stage taps have the right geometry too
            eng.enable_taps(True)
            model(x.to(DEV))
            for n in ("stem", "s2", "tok0", "blk11", "l4", "p1", "h1"):
                got, want = eng.tap(n), otaps[n]
                assert got.shape == want.shape, (n, got.shape, want.shape)
                rel = ((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
                assert rel < STAGE_RMS_REL["bf16"], (n, rel)


def test_growing_the_arena_keeps_384_bit_identical():
    """A model that has re-planned its arena for a larger input returns the very same bits at 384x384."""
    sd = random_state_dict(0, 3)
    x = synthetic_input(5, 4, "normal").to(DEV)
    fresh = make_model(sd, 3, "bf16", 4)
    y0 = fresh(x).clone()
    grown = make_model(sd, 3, "bf16", 4)
    big = synthetic_input(9, 2, "normal", (512, 640)).to(DEV)
    yb = grown(big)
    assert yb.shape == (2, 3, 512, 640) and torch.isfinite(yb).all()
    assert grown.max_hw == (512, 640)
    assert torch.equal(grown(x), y0)
    # batch invariance at a non-native size
    assert torch.equal(grown(big[1:2])[0], yb[1])


def test_flex_rejects_bad_sizes():
    sd = random_state_dict(0, 1)
    m = make_model(sd, 1, "bf16", 1)
    with pytest.raises(ValueError):
        m(torch.zeros(1, 3, 384, 400, device=DEV))
    eng = m._get_engine(torch.device(DEV))
    with pytest.raises(RuntimeError, match="larger than the engine was planned for"):
        eng.forward(torch.zeros(1, 3, 416, 384, device=DEV))
    with pytest.raises(RuntimeError, match="multiples of 32"):
        eng.forward(torch.zeros(1, 3, 32, 384, device=DEV))
