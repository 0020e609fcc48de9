"""Pins oracle/dpt_oracle.py against the reference's OWN modules and emits golden vectors.

Runs only where /root/reference exists (the build container).  It
  1. installs oracle/timm_shim.py as ``timm`` and imports the reference's unmodified
     ``modules/midas/dpt_depth.py`` (+ blocks.py, vit.py) from
     /root/reference/omnidata_tools/torch;
  2. checks that ``omnidata_amd.weights.state_dict_spec`` equals the reference model's
     state_dict keys/shapes exactly (strict load works);
  3. runs reference-wired model vs the functional oracle on identical seeded weights and
     inputs and reports max-abs differences (must be ~1e-6: fp32 reassociation only);
  4. writes tests/golden/dpt_{task}_seed{S}.npz: sub-sampled reference outputs + stage
     taps + whole-tensor statistics.  The weights/inputs are regenerated from the seed at
     test time (omnidata_amd.weights.random_state_dict / synthetic_input), so fixtures stay
     small.

Usage: python -m oracle.validate_vs_reference [--no-write]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/omnidata_tools/torch"

from oracle import timm_shim  # noqa: E402
from oracle.dpt_oracle import dpt_forward, dpt_forward_vitl16  # noqa: E402
from omnidata_amd.weights import random_state_dict, state_dict_spec, synthetic_input  # noqa: E402

GOLDEN_TAPS = ("stem", "s0", "s1", "s2", "tok0", "blk0", "blk8", "blk11", "l3", "l4",
               "l1_rn", "l4_rn", "p4", "p3", "p2", "p1", "h0", "h1", "pre")
CASES = [("normal", 3, 0, 1), ("normal", 3, 1, 2), ("depth", 1, 0, 1), ("depth", 1, 2, 1)]
# inputs that are not 384x384 exercise forward_flex's pos_embed resize (vit.py:102-125); written as flex_*.npz
FLEX_CASES = [("normal", 3, 3, 1, (256, 320)), ("depth", 1, 4, 2, (448, 288))]


def subsample(t: torch.Tensor) -> np.ndarray:
    """Deterministic small view of a tap: every 7th element of the flattened tensor,
    capped at 4096 values."""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // 4096)
    return f[::step][:4096].double().numpy().astype(np.float32)


def stats(t: torch.Tensor) -> np.ndarray:
    d = t.detach().double()
    return np.array([d.mean().item(), d.std().item(), d.abs().max().item(), d.sum().item()], dtype=np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-write", action="store_true")
    args = ap.parse_args()
    assert os.path.isdir(REF), "reference checkout not present"
    timm_shim.install()
    sys.path.insert(0, REF)
    from modules.midas.dpt_depth import DPTDepthModel  # the reference's own class
    import modules.midas.vit as ref_vit

    torch.set_num_threads(os.cpu_count())
    worst = 0.0
    for case in CASES + FLEX_CASES:
        task, C, seed, B = case[:4]
        H, W = case[4] if len(case) > 4 else (384, 384)
        model = DPTDepthModel(backbone="vitb_rn50_384", num_channels=C).eval()
        ref_sd = model.state_dict()
        spec = state_dict_spec(C)
        assert list(ref_sd.keys()) == list(spec.keys()) or set(ref_sd.keys()) == set(spec.keys()), (
            set(ref_sd.keys()) ^ set(spec.keys()))
        for k, shp in spec.items():
            assert tuple(ref_sd[k].shape) == tuple(shp), (k, ref_sd[k].shape, shp)
        sd = random_state_dict(seed, C)
        model.load_state_dict(sd, strict=True)
        x = synthetic_input(seed, B, task, (H, W))
        with torch.no_grad():
            y_ref = model(x)
        # stage taps of the reference run: hooks dict + re-run pieces through modules
        acts = ref_vit.activations
        taps = {}
        y_or = dpt_forward(sd, x, taps)
        d_out = (y_ref - y_or).abs().max().item()
        d_l1 = (acts["1"] - taps["s0"]).abs().max().item()
        d_l2 = (acts["2"] - taps["s1"]).abs().max().item()
        d_b8 = (acts["3"] - taps["blk8"]).abs().max().item()
        d_b11 = (acts["4"] - taps["blk11"]).abs().max().item()
        worst = max(worst, d_out)
        print(f"[{task} seed={seed} B={B} {H}x{W}] ref-vs-oracle max|d|: out={d_out:.3e} hook1={d_l1:.3e} "
              f"hook2={d_l2:.3e} blk8={d_b8:.3e} blk11={d_b11:.3e}; out mean={y_ref.mean():.4f} "
              f"std={y_ref.std():.4f} min={y_ref.min():.4f} max={y_ref.max():.4f} "
              f"frac0={(y_ref == 0).float().mean():.4f} frac>1={(y_ref > 1).float().mean():.4f}")
        for name in ("s0", "s2", "tok0", "blk11", "l3", "l4", "p4", "p1", "h0", "h1", "pre"):
            t = taps[name]
            print(f"    tap {name:6s} shape={tuple(t.shape)} mean={t.mean():+.3f} std={t.std():.3f} absmax={t.abs().max():.2f}")
        assert y_ref.shape == ((B, 3, H, W) if C == 3 else (B, H, W)), y_ref.shape
        assert d_out < 2e-4, d_out
        if not args.no_write:
            out = {"task": task, "num_channels": C, "seed": seed, "batch": B,
                   "out_sub": subsample(y_ref), "out_stats": stats(y_ref),
                   "height": H, "width": W,
                   "out_row": y_ref.reshape(B, -1, H, W)[0, 0, H // 2 - 1].numpy().astype(np.float32)}
            for name in GOLDEN_TAPS:
                out["tap_" + name] = subsample(taps[name])
                out["stat_" + name] = stats(taps[name])
            fname = f"dpt_{task}_seed{seed}.npz" if (H, W) == (384, 384) else f"flex_{task}_seed{seed}_{H}x{W}.npz"
            path = os.path.join(ROOT, "tests", "golden", fname)
            np.savez_compressed(path, **out)
            print("    wrote", path, os.path.getsize(path), "bytes")
    # ---- DPT-Large (backbone='vitl16_384', demo.py:81): one depth case -> tests/golden/vitl16_depth_seed5.npz
    VITL_TAPS = ("tok0", "blk5", "blk23", "l1", "l2", "l3", "l4", "l1_rn", "l4_rn", "p4", "p1", "h0", "h1", "pre")
    for task, C, seed, B in [("depth", 1, 5, 1)]:
        model = DPTDepthModel(backbone="vitl16_384", num_channels=C).eval()
        ref_sd = model.state_dict()
        spec = state_dict_spec(C, True, "vitl16_384")
        assert set(ref_sd.keys()) == set(spec.keys()), set(ref_sd.keys()) ^ set(spec.keys())
        for k, shp in spec.items():
            assert tuple(ref_sd[k].shape) == tuple(shp), (k, ref_sd[k].shape, shp)
        sd = random_state_dict(seed, C, backbone="vitl16_384")
        model.load_state_dict(sd, strict=True)
        x = synthetic_input(seed, B, task)
        with torch.no_grad():
            y_ref = model(x)
        acts = ref_vit.activations
        taps = {}
        y_or = dpt_forward_vitl16(sd, x, taps)
        d_out = (y_ref - y_or).abs().max().item()
        d_h = [(acts[str(i + 1)] - taps[f"blk{l}"]).abs().max().item() for i, l in enumerate((5, 11, 17, 23))]
        worst = max(worst, d_out)
        print(f"[vitl16_384 {task} seed={seed} B={B}] ref-vs-oracle max|d|: out={d_out:.3e} hooks={['%.2e' % v for v in d_h]}; "
              f"out mean={y_ref.mean():.4f} std={y_ref.std():.4f} min={y_ref.min():.4f} max={y_ref.max():.4f} "
              f"frac0={(y_ref == 0).float().mean():.4f}")
        for name in VITL_TAPS:
            t = taps[name]
            print(f"    tap {name:6s} shape={tuple(t.shape)} mean={t.mean():+.3f} std={t.std():.3f} absmax={t.abs().max():.2f}")
        assert d_out < 2e-4, d_out
        if not args.no_write:
            out = {"task": task, "num_channels": C, "seed": seed, "batch": B, "backbone": "vitl16_384",
                   "out_sub": subsample(y_ref), "out_stats": stats(y_ref), "height": 384, "width": 384,
                   "out_row": y_ref.reshape(B, -1, 384, 384)[0, 0, 191].numpy().astype(np.float32)}
            for name in VITL_TAPS:
                out["tap_" + name] = subsample(taps[name])
                out["stat_" + name] = stats(taps[name])
            path = os.path.join(ROOT, "tests", "golden", f"vitl16_{task}_seed{seed}.npz")
            np.savez_compressed(path, **out)
            print("    wrote", path, os.path.getsize(path), "bytes")
    print("worst ref-vs-oracle output difference:", worst)


if __name__ == "__main__":
    main()
