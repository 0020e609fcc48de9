#!/usr/bin/env python3
"""Real-checkpoint harness: everything that can be said about a checkpoint in ONE run.

    python tools/eval_checkpoint.py --task normal --ckpt pretrained_models/omnidata_dpt_normal_v2.ckpt --images DIR [--gt DIR]
                                    [--dtypes mixed,bf16x3] [--batch 8] [--out report.json]

What the reference does with a checkpoint is `demo.py:62-72` (load, strip the Lightning prefix, strict load_state_dict) and
`paper_code/evaluation_metrics.py:14-106` (the metrics of docs/pretrained.html:96-104).  This tool loads the same file through
`omnidata_amd.weights.read_checkpoint`, preprocesses the images exactly as demo.py does (`omnidata_amd.preprocess`), and for
every requested dtype reports

  * the fp16 RANGE FLAG (include/dptx.h dptx_range_status) -- does any activation leave the fp16 range on THESE weights?
    (answers VERDICT r5 W2 for a real checkpoint in one run; the models' overflow fallback is switched OFF here so that the
    flag is seen, not acted on),
  * max |activation| at every stage tap of the first batch (stem, s0..s2, l3, l4, l*_rn, p4..p1, h0): how far from 65504,
  * the output's range and its agreement with the first dtype of the list (max / rms difference; `bf16x3` and `fp16x3` are
    reference-grade, 1e-4 / 1e-5 from the fp32 forward on the synthetic families, so `mixed` vs `bf16x3` is the parity
    check that needs no CPU reference),
  * with --gt: `omnidata_amd.metrics.get_metrics` (angular error / within-t degrees for normals; L1, log10, rel, irmse, si_log
    for depth) against ground truth.  Ground-truth files: <gt>/<stem>.npy or <stem>.png for every image <stem>.*; normals as
    [H,W,3] in [0,1] (the reference's (n+1)/2 convention; an .npy in [-1,1] is mapped), depth as [H,W] in [0,1]; optional
    validity mask <stem>_mask.png (non-zero = valid).  They are resized / centre-cropped like the input.

There is no CPU fallback: the forward runs on an MI355X through libdptx.so.  No network is needed or used.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

TAPS = ["stem", "s0", "s1", "s2", "l3", "l4", "l1_rn", "l2_rn", "l3_rn", "l4_rn", "p4", "p3", "p2", "p1", "h0"]
IMG_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".webp")


def load_gt(gt_dir: str, stem: str, task: str):
    """-> (target [C,384,384] float tensor, mask [1,384,384] bool tensor) or None."""
    from omnidata_amd import preprocess as pp
    arr = None
    for ext in (".npy", ".png"):
        p = os.path.join(gt_dir, stem + ext)
        if os.path.exists(p):
            arr = np.load(p).astype(np.float32) if ext == ".npy" else np.asarray(Image.open(p)).astype(np.float32) / 255.0
            break
    if arr is None:
        return None
    if task == "normal":
        if arr.ndim != 3 or arr.shape[2] < 3:
            raise ValueError(f"{stem}: normal ground truth must be [H,W,3]")
        arr = arr[..., :3]
        if arr.min() < -0.01:          # [-1,1] vectors -> the reference's [0,1] image convention
            arr = (arr + 1.0) * 0.5
    else:
        arr = arr[..., 0] if arr.ndim == 3 else arr
        arr = arr[..., None]
    mpath = os.path.join(gt_dir, stem + "_mask.png")
    mask = (np.asarray(Image.open(mpath).convert("L")) > 0) if os.path.exists(mpath) else np.ones(arr.shape[:2], bool)

    def fit(a, nearest):   # same geometry as the input: Resize(384, shorter side) + CenterCrop(384)
        chans = []
        for c in range(a.shape[2]):
            im = Image.fromarray(a[..., c].astype(np.float32), mode="F")
            w, h = im.size
            s = 384.0 / min(w, h)
            im = im.resize((max(384, round(w * s)), max(384, round(h * s))), Image.NEAREST if nearest else Image.BILINEAR)
            chans.append(np.asarray(pp.center_crop(im, 384), dtype=np.float32))
        return torch.from_numpy(np.stack(chans))
    return fit(arr, False), fit(mask[..., None].astype(np.float32), True) > 0.5


def evaluate(task: str, ckpt: str, images, gt_dir=None, dtypes=("mixed", "bf16x3"), batch: int = 8, backbone: str = "vitb_rn50_384",
             device: str = "cuda:0") -> dict:
    from omnidata_amd import preprocess as pp
    from omnidata_amd.metrics import get_metrics
    from omnidata_amd.model import build_model
    from omnidata_amd.weights import read_checkpoint
    if not torch.cuda.is_available():
        raise RuntimeError("eval_checkpoint needs an AMD GPU: the DPT forward exists only as HIP kernels (no CPU fallback)")
    sd = read_checkpoint(ckpt)
    dev = torch.device(device)
    stems = [os.path.splitext(os.path.basename(p))[0] for p in images]
    xs = torch.cat([pp.image_to_input_gpu(Image.open(p), task, dev) for p in images])     # demo.py:74-76 / 92-95 on the GPU
    report = {"checkpoint": os.path.abspath(ckpt), "task": task, "backbone": backbone, "images": len(images), "tensors": len(sd),
              "parameters": int(sum(v.numel() for v in sd.values())), "dtypes": {}}
    ref_out = None
    for dt in dtypes:
        model = build_model(task, backbone=backbone, dtype=dt, max_batch=min(batch, 48), overflow_fallback=False)
        model.load_state_dict(sd)          # strict, like demo.py:72
        model.to(dev)
        outs = []
        taps = {}
        for i in range(0, xs.shape[0], batch):
            xb = xs[i:i + batch]
            if i == 0:   # stage taps of the first batch: how large do the activations of THESE weights get
                eng = model._get_engine(dev)
                eng.enable_taps(True)
                model(xb)
                for name in TAPS:
                    try:
                        t = eng.tap(name)
                        taps[name] = {"max_abs": float(t.abs().max()), "finite": bool(torch.isfinite(t).all())}
                    except RuntimeError:
                        pass   # not a tap of this backbone
                eng.enable_taps(False)
            outs.append(model(xb).float())
        y = torch.cat(outs)
        y = y[:, None] if y.dim() == 3 else y
        eng = model.engine
        entry = {"range_flag_overflow": bool(eng.range_overflowed(reset=True)) if dt in ("mixed", "fp16x3", "fp16") else None,
                 "fp16_max": 65504.0, "stage_tap_max_abs": taps,
                 "largest_tap": max(taps.items(), key=lambda kv: kv[1]["max_abs"])[0] if taps else None,
                 "output": {"min": float(y.min()), "max": float(y.max()), "mean": float(y.mean()), "finite": bool(torch.isfinite(y).all())}}
        if ref_out is None:
            ref_out = y
        else:
            d = (y - ref_out).abs()
            entry["vs_" + dtypes[0]] = {"max_abs": float(d.max()), "rms": float(d.pow(2).mean().sqrt()), "meets_1e-3": bool(d.max() < 1e-3)}
        if gt_dir:
            pairs = [(k, load_gt(gt_dir, s, task)) for k, s in enumerate(stems)]
            pairs = [(k, g) for k, g in pairs if g is not None]
            if pairs:
                idx = [k for k, _ in pairs]
                tgt = torch.stack([g[0] for _, g in pairs]).to(dev)
                msk = torch.stack([g[1] for _, g in pairs]).to(dev)
                pred = y[idx].clamp(0, 1)   # demo.py:140
                entry["metrics"] = get_metrics(pred, tgt, "normal" if task == "normal" else "depth_zbuffer", msk)
                entry["metrics_images"] = len(idx)
        report["dtypes"][dt] = entry
        del model
        torch.cuda.empty_cache()
    return report


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--task", required=True, choices=["normal", "depth"])
    ap.add_argument("--ckpt", required=True, help="Lightning .ckpt / raw state_dict / MiDaS {'optimizer','model'} file (demo.py:62-72)")
    ap.add_argument("--images", required=True, help="image file or directory (demo.py:158-160 iterates glob(dir + '/*'))")
    ap.add_argument("--gt", default=None, help="ground-truth directory (see the module docstring)")
    ap.add_argument("--dtypes", default="bf16x3,mixed", help="first one is the reference the others are compared with")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--backbone", default="vitb_rn50_384", choices=["vitb_rn50_384", "vitl16_384"])
    ap.add_argument("--out", default=None)
    args = ap.parse_args(argv)
    if os.path.isdir(args.images):
        images = sorted(p for p in glob.glob(os.path.join(args.images, "*")) if p.lower().endswith(IMG_EXT) and not p.endswith("_mask.png"))
    else:
        images = [args.images]
    if not images or not all(os.path.isfile(p) for p in images):
        print("invalid file path!")   # demo.py:161-163
        sys.exit(1)
    rep = evaluate(args.task, args.ckpt, images, args.gt, tuple(d for d in args.dtypes.split(",") if d), args.batch, args.backbone)
    txt = json.dumps(rep, indent=1)
    if args.out:
        with open(args.out, "w") as f:
            f.write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
