"""Pins omnidata_amd/metrics.py against the reference's paper_code/evaluation_metrics.py:get_metrics and writes
tests/golden/metrics_seed*.npz (inputs are regenerated from the seed; the reference's numbers are stored).

Runs only where /root/reference exists.  The reference module imports torchvision, pandas and tqdm at the top without
using them in get_metrics; missing ones are stubbed so that the file imports unmodified.
Usage: python -m oracle.validate_metrics_vs_reference
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/paper_code/evaluation_metrics.py"


def make_case(seed: int, task: str):
    g = torch.Generator().manual_seed(4242 + seed)
    B, H, W = 2, 24, 32
    if task == "normal":
        t = torch.randn(B, 3, H, W, generator=g)
        t = t / t.norm(dim=1, keepdim=True)
        p = t + 0.35 * torch.randn(B, 3, H, W, generator=g)
        t, p = t * 0.5 + 0.5, p * 0.5 + 0.5  # the [0,1] encoding the models output
    else:
        t = torch.rand(B, 1, H, W, generator=g) * 0.9 + 0.05
        p = (t + 0.05 * torch.randn(B, 1, H, W, generator=g)).clamp(min=0.0)
    m = torch.rand(B, 1, H, W, generator=g) > 0.3
    return p, t, m


def load_reference():
    for name in ("torchvision", "pandas", "tqdm"):
        try:
            __import__(name)
        except Exception:
            mod = types.ModuleType(name)
            if name == "tqdm":
                mod.tqdm = lambda x, *a, **k: x
            sys.modules[name] = mod
    spec = importlib.util.spec_from_file_location("ref_evaluation_metrics", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    assert os.path.isfile(REF), "reference checkout not present"
    ref = load_reference()
    from omnidata_amd.metrics import get_metrics
    worst = 0.0
    for seed in (0, 1):
        for task in ("normal", "depth_zbuffer"):
            p, t, m = make_case(seed, task)
            want = ref.get_metrics(p, t, task=task, masks=m.expand(-1, p.shape[1], -1, -1))
            got = get_metrics(p, t, task=task, masks=m)
            out = {}
            for k, v in want.items():
                v = float(v)
                d = abs(v - got[k]) / max(1.0, abs(v))
                worst = max(worst, d)
                assert d < 1e-6, (task, k, v, got[k])  # float64 both sides; only summation order differs
                out[k] = v
            path = os.path.join(ROOT, "tests", "golden", f"metrics_{task}_seed{seed}.npz")
            np.savez(path, keys=np.array(list(out.keys())), values=np.array(list(out.values()), dtype=np.float64))
            print(task, seed, {k: round(v, 5) for k, v in out.items()})
    print("worst relative difference to the reference:", worst)


if __name__ == "__main__":
    main()
