"""The CPU oracle against the golden vectors produced by the REFERENCE's own modules
(oracle/validate_vs_reference.py, run where /root/reference exists).  No GPU."""
import glob
import os

import numpy as np
import pytest
import torch

from omnidata_amd.weights import random_state_dict, synthetic_input
from oracle.dpt_oracle import dpt_forward, ssi_align, mean_angular_error_deg
from oracle.validate_vs_reference import GOLDEN_TAPS, stats, subsample

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "dpt_*.npz")))
GOLDEN += sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "flex_*.npz")))  # non-384 inputs


def test_golden_files_present():
    assert len(GOLDEN) >= 6


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_matches_reference_golden(path):
    g = np.load(path)
    task, C, seed, B = str(g["task"]), int(g["num_channels"]), int(g["seed"]), int(g["batch"])
    H, W = (int(g["height"]), int(g["width"])) if "height" in g else (384, 384)
    torch.set_num_threads(os.cpu_count())
    taps = {}
    y = dpt_forward(random_state_dict(seed, C), synthetic_input(seed, B, task, (H, W)), taps)
    assert tuple(y.shape) == ((B, 3, H, W) if C == 3 else (B, H, W))
    # fp32 oracle vs fp32 reference: same ops, so only thread-count reassociation noise
    np.testing.assert_allclose(subsample(y), g["out_sub"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(y.reshape(B, -1, H, W)[0, 0, H // 2 - 1].numpy(), g["out_row"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(stats(y)[:3], g["out_stats"][:3], rtol=1e-4, atol=1e-5)
    for name in GOLDEN_TAPS:
        ref = g["tap_" + name]
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(subsample(taps[name]), ref, atol=3e-5 * scale, rtol=0, err_msg=name)


def test_vitl16_oracle_matches_reference_golden():
    """DPT-Large (backbone='vitl16_384'): the functional oracle against the vector produced by the reference's own
    DPTDepthModel(backbone='vitl16_384') (oracle/validate_vs_reference.py)."""
    from oracle.dpt_oracle import dpt_forward_vitl16
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vitl16_depth_seed5.npz"))
    seed, B = int(g["seed"]), int(g["batch"])
    torch.set_num_threads(os.cpu_count())
    taps = {}
    y = dpt_forward_vitl16(random_state_dict(seed, 1, backbone="vitl16_384"), synthetic_input(seed, B, "depth"), taps)
    assert tuple(y.shape) == (B, 384, 384)
    np.testing.assert_allclose(subsample(y), g["out_sub"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(y.reshape(B, -1, 384, 384)[0, 0, 191].numpy(), g["out_row"], atol=2e-5, rtol=0)
    for name in ("tok0", "blk5", "blk23", "l1", "l2", "l3", "l4", "p4", "p1", "h0", "pre"):
        ref = g["tap_" + name]
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(subsample(taps[name]), ref, atol=3e-5 * scale, rtol=0, err_msg=name)


def test_metrics_helpers():
    a = torch.rand(2, 16, 16)
    assert torch.allclose(ssi_align(3 * a + 0.5, a), a, atol=1e-5)
    n = torch.rand(1, 3, 8, 8)
    assert mean_angular_error_deg(n, n) < 0.05
