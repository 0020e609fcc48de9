"""Evaluation metrics of the omnidata paper for surface normals and z-buffer depth, restated for torch tensors on any
device (definitions follow paper_code/evaluation_metrics.py:13-106 of the reference, which is what
produced docs/pretrained.html's tables).  Evaluation-side code: nothing here is on the forward's hot path.

    normal_metrics(pred, target, mask)  ->  dict(ang_error_mean, ang_error_median, percentage_within_{11.25,22.5,30}_degrees,
                                                 ang_error_without_masking, eval_L1, eval_mse)
    depth_metrics(pred, target, mask)   ->  dict(eval_L1, eval_mse, log10_diff, log10, rel_error, irmse, si_log)

pred / target: [B, C, H, W]; mask: [B, 1 or C, H, W] bool (only channel 0 is used, as in the reference).
Everything is accumulated in float64.  Returns None when the mask is empty.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

__all__ = ["normal_metrics", "depth_metrics", "get_metrics"]


def _flatten(pred: torch.Tensor, target: torch.Tensor, mask: torch.Tensor):
    p = pred.detach().permute(0, 2, 3, 1).double()
    t = target.detach().permute(0, 2, 3, 1).double()
    m = mask.detach().permute(0, 2, 3, 1)[..., 0].bool()
    return p.reshape(-1, p.shape[-1]), t.reshape(-1, t.shape[-1]), m


def _l1_mse(diff: torch.Tensor, inv_valid: float) -> Dict[str, float]:
    # means run over ALL pixels and channels and are rescaled by numel/valid, as the reference does
    return {"eval_L1": float(diff.mean() * inv_valid * 100.0), "eval_mse": float((diff * diff).mean() * inv_valid * 100.0)}


def normal_metrics(pred: torch.Tensor, target: torch.Tensor, mask: torch.Tensor) -> Optional[Dict[str, float]]:
    p, t, m = _flatten(pred, target, mask)
    n_valid = int(m.sum())
    if n_valid < 1:
        return None
    fm = m.reshape(-1)
    inv_valid = fm.numel() / n_valid
    np_, nt = p.norm(dim=1), t.norm(dim=1)
    cos = ((p * t).sum(dim=1) / (np_ * nt).clamp(min=1e-8)).clamp(-1.0, 1.0)
    ang = torch.acos(cos) * (180.0 / math.pi)
    ang_valid = ang[fm]
    # the reference takes numpy's median (mean of the two middle values for an even count) of the valid pixels
    srt = ang_valid.sort().values
    k = srt.numel()
    median = float(srt[k // 2]) if k % 2 else float((srt[k // 2 - 1] + srt[k // 2]) * 0.5)
    out = {
        "ang_error_mean": float((ang * fm).sum() / n_valid),
        "ang_error_median": median,
        "ang_error_without_masking": float(ang.mean()),
        "percentage_within_11.25_degrees": float((ang_valid <= 11.25).sum().double() / n_valid),
        "percentage_within_22.5_degrees": float((ang_valid <= 22.5).sum().double() / n_valid),
        "percentage_within_30_degrees": float((ang_valid <= 30.0).sum().double() / n_valid),
    }
    # L1 / MSE are taken between softly normalised vectors (norm + 2e-2)
    diff = ((p / (np_[:, None] + 2e-2)) - (t / (nt[:, None] + 2e-2))).abs() * fm[:, None]
    out.update(_l1_mse(diff, inv_valid))
    return out


def depth_metrics(pred: torch.Tensor, target: torch.Tensor, mask: torch.Tensor) -> Optional[Dict[str, float]]:
    p, t, m = _flatten(pred, target, mask)
    n_valid = int(m.sum())
    if n_valid < 1:
        return None
    fm = m.reshape(-1)[:, None]
    inv_valid = fm.numel() / n_valid
    diff = (p - t).abs() * fm
    out = _l1_mse(diff, inv_valid)
    lp, lt = torch.log(1.0 + 64.0 * p), torch.log(1.0 + 64.0 * t)
    dlog = ((lp - lt) * fm).abs()
    out["log10_diff"] = float((torch.log(1.0 + 64.0 * diff) * fm).mean() * inv_valid)
    out["log10"] = float(dlog.mean() * inv_valid)
    out["si_log"] = float((dlog ** 2).sum() / n_valid - dlog.sum() ** 2 / (n_valid ** 2))
    out["rel_error"] = float(((diff / t) * fm).mean() * inv_valid)
    out["irmse"] = float((((1.0 / (1.0 + 64.0 * p) - 1.0 / (1.0 + 64.0 * t)) ** 2) * fm).mean() * inv_valid)
    return out


def get_metrics(pred, target, task=None, masks=None):
    """Reference-shaped entry point (evaluation_metrics.py:13): task in {'normal', 'depth_zbuffer'}."""
    if task == "normal":
        return normal_metrics(pred, target, masks)
    if task == "depth_zbuffer":
        return depth_metrics(pred, target, masks)
    p, t, m = _flatten(pred, target, masks)
    n_valid = int(m.sum())
    if n_valid < 1:
        return None
    return _l1_mse((p - t).abs() * m.reshape(-1)[:, None], m.numel() / n_valid)
