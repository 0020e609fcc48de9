"""State-dict specification, checkpoint reading and seeded synthetic weights.

Key names / shapes are the reference's (after stripping Lightning's ``model.`` prefix,
``omnidata_tools/torch/demo.py:65-70``): the reference-side modules of
``modules/midas/dpt_depth.py`` + ``blocks.py`` + ``vit.py`` and, under
``pretrained.model.``, the timm ``vit_base_resnet50_384`` backbone
(``modules/midas/vit.py:483``).  See SURVEY.md A.3.

There is no network in the build environment, so the published checkpoints
(``tools/download_surface_normal_models.sh:23-25``) cannot be fetched; parity and
benchmarks use ``random_state_dict(seed, num_channels)`` whose scales keep every
activation O(1) through the 12 transformer blocks and ~60 convolutions and centre the
head output in (0, 1) so that the final ReLU / clamp do not hide errors.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import torch

STAGE_DEPTHS = (3, 4, 9)
STAGE_OUT = (256, 512, 1024)
VIT_DEPTH, VIT_DIM, VIT_MLP = 12, 768, 3072
FEATURES = 256


BACKBONES = ("vitb_rn50_384", "vitl16_384")


def vitl16_state_dict_spec(num_channels: int = 1, include_unused: bool = True) -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered {key: shape} of ``DPTDepthModel(backbone='vitl16_384', num_channels=C)`` (DPT-Large: blocks.py:12-18,
    vit.py:299-309 hooks [5,11,17,23], reassemble vit.py:176-260): timm ``vit_large_patch16_384`` under ``pretrained.model.``
    (16x16 patch conv, 24 blocks of width 1024, 16 heads), four ProjectReadouts, ConvTranspose2d up-sampling in
    act_postprocess1/2, and the same scratch / RefineNet / head modules with layerN_rn inputs [256, 512, 1024, 1024]."""
    D, depth, mlp = 1024, 24, 4096
    sp: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    vp = "pretrained.model."
    sp[vp + "cls_token"] = (1, 1, D)
    sp[vp + "pos_embed"] = (1, 577, D)
    sp[vp + "patch_embed.proj.weight"] = (D, 3, 16, 16)
    sp[vp + "patch_embed.proj.bias"] = (D,)
    for l in range(depth):
        p = f"{vp}blocks.{l}."
        sp[p + "norm1.weight"] = (D,)
        sp[p + "norm1.bias"] = (D,)
        sp[p + "attn.qkv.weight"] = (3 * D, D)
        sp[p + "attn.qkv.bias"] = (3 * D,)
        sp[p + "attn.proj.weight"] = (D, D)
        sp[p + "attn.proj.bias"] = (D,)
        sp[p + "norm2.weight"] = (D,)
        sp[p + "norm2.bias"] = (D,)
        sp[p + "mlp.fc1.weight"] = (mlp, D)
        sp[p + "mlp.fc1.bias"] = (mlp,)
        sp[p + "mlp.fc2.weight"] = (D, mlp)
        sp[p + "mlp.fc2.bias"] = (D,)
    if include_unused:
        sp[vp + "norm.weight"] = (D,)
        sp[vp + "norm.bias"] = (D,)
        sp[vp + "head.weight"] = (1000, D)
        sp[vp + "head.bias"] = (1000,)
    feats = (256, 512, 1024, 1024)
    for n, f in enumerate(feats, start=1):
        p = f"pretrained.act_postprocess{n}."
        sp[p + "0.project.0.weight"] = (D, 2 * D)
        sp[p + "0.project.0.bias"] = (D,)
        sp[p + "3.weight"] = (f, D, 1, 1)
        sp[p + "3.bias"] = (f,)
        if n == 1:
            sp[p + "4.weight"] = (f, f, 4, 4)   # ConvTranspose2d(k=4, s=4): [Cin, Cout, kh, kw]
            sp[p + "4.bias"] = (f,)
        elif n == 2:
            sp[p + "4.weight"] = (f, f, 2, 2)   # ConvTranspose2d(k=2, s=2)
            sp[p + "4.bias"] = (f,)
        elif n == 4:
            sp[p + "4.weight"] = (f, f, 3, 3)   # Conv2d(k=3, s=2, p=1)
            sp[p + "4.bias"] = (f,)
    for i, c in enumerate(feats, start=1):
        sp[f"scratch.layer{i}_rn.weight"] = (FEATURES, c, 3, 3)
    for i in (1, 2, 3, 4):
        p = f"scratch.refinenet{i}."
        sp[p + "out_conv.weight"] = (FEATURES, FEATURES, 1, 1)
        sp[p + "out_conv.bias"] = (FEATURES,)
        for u in (1, 2):
            if u == 1 and i == 4 and not include_unused:
                continue
            for c in (1, 2):
                sp[f"{p}resConfUnit{u}.conv{c}.weight"] = (FEATURES, FEATURES, 3, 3)
                sp[f"{p}resConfUnit{u}.conv{c}.bias"] = (FEATURES,)
    oc = "scratch.output_conv."
    sp[oc + "0.weight"] = (FEATURES // 2, FEATURES, 3, 3)
    sp[oc + "0.bias"] = (FEATURES // 2,)
    sp[oc + "2.weight"] = (32, FEATURES // 2, 3, 3)
    sp[oc + "2.bias"] = (32,)
    sp[oc + "4.weight"] = (num_channels, 32, 1, 1)
    sp[oc + "4.bias"] = (num_channels,)
    return sp


def state_dict_spec(num_channels: int = 3, include_unused: bool = True,
                    backbone: str = "vitb_rn50_384") -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered {key: shape} of ``DPTDepthModel(backbone='vitb_rn50_384', num_channels=C)``.

    ``include_unused`` adds the tensors a real checkpoint carries but the forward never
    reads: timm's final ``norm`` + 1000-class ``head`` and ``refinenet4.resConfUnit1``
    (blocks.py:329-333 only runs resConfUnit1 when two inputs are given).
    """
    if backbone == "vitl16_384":
        return vitl16_state_dict_spec(num_channels, include_unused)
    if backbone != "vitb_rn50_384":
        raise ValueError(f"backbone must be one of {BACKBONES}")
    sp: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    vp = "pretrained.model."
    sp[vp + "cls_token"] = (1, 1, VIT_DIM)
    sp[vp + "pos_embed"] = (1, 577, VIT_DIM)
    bp = vp + "patch_embed.backbone."
    sp[bp + "stem.conv.weight"] = (64, 3, 7, 7)
    sp[bp + "stem.norm.weight"] = (64,)
    sp[bp + "stem.norm.bias"] = (64,)
    cin = 64
    for s, (depth, cout) in enumerate(zip(STAGE_DEPTHS, STAGE_OUT)):
        mid = cout // 4
        for b in range(depth):
            p = f"{bp}stages.{s}.blocks.{b}."
            if b == 0:
                sp[p + "downsample.conv.weight"] = (cout, cin, 1, 1)
                sp[p + "downsample.norm.weight"] = (cout,)
                sp[p + "downsample.norm.bias"] = (cout,)
            sp[p + "conv1.weight"] = (mid, cin, 1, 1)
            sp[p + "norm1.weight"] = (mid,)
            sp[p + "norm1.bias"] = (mid,)
            sp[p + "conv2.weight"] = (mid, mid, 3, 3)
            sp[p + "norm2.weight"] = (mid,)
            sp[p + "norm2.bias"] = (mid,)
            sp[p + "conv3.weight"] = (cout, mid, 1, 1)
            sp[p + "norm3.weight"] = (cout,)
            sp[p + "norm3.bias"] = (cout,)
            cin = cout
    sp[vp + "patch_embed.proj.weight"] = (VIT_DIM, 1024, 1, 1)
    sp[vp + "patch_embed.proj.bias"] = (VIT_DIM,)
    for l in range(VIT_DEPTH):
        p = f"{vp}blocks.{l}."
        sp[p + "norm1.weight"] = (VIT_DIM,)
        sp[p + "norm1.bias"] = (VIT_DIM,)
        sp[p + "attn.qkv.weight"] = (3 * VIT_DIM, VIT_DIM)
        sp[p + "attn.qkv.bias"] = (3 * VIT_DIM,)
        sp[p + "attn.proj.weight"] = (VIT_DIM, VIT_DIM)
        sp[p + "attn.proj.bias"] = (VIT_DIM,)
        sp[p + "norm2.weight"] = (VIT_DIM,)
        sp[p + "norm2.bias"] = (VIT_DIM,)
        sp[p + "mlp.fc1.weight"] = (VIT_MLP, VIT_DIM)
        sp[p + "mlp.fc1.bias"] = (VIT_MLP,)
        sp[p + "mlp.fc2.weight"] = (VIT_DIM, VIT_MLP)
        sp[p + "mlp.fc2.bias"] = (VIT_DIM,)
    if include_unused:
        sp[vp + "norm.weight"] = (VIT_DIM,)
        sp[vp + "norm.bias"] = (VIT_DIM,)
        sp[vp + "head.weight"] = (1000, VIT_DIM)
        sp[vp + "head.bias"] = (1000,)
    for n in (3, 4):
        p = f"pretrained.act_postprocess{n}."
        sp[p + "0.project.0.weight"] = (VIT_DIM, 2 * VIT_DIM)
        sp[p + "0.project.0.bias"] = (VIT_DIM,)
        sp[p + "3.weight"] = (VIT_DIM, VIT_DIM, 1, 1)
        sp[p + "3.bias"] = (VIT_DIM,)
        if n == 4:
            sp[p + "4.weight"] = (VIT_DIM, VIT_DIM, 3, 3)
            sp[p + "4.bias"] = (VIT_DIM,)
    for i, c in enumerate((256, 512, 768, 768), start=1):
        sp[f"scratch.layer{i}_rn.weight"] = (FEATURES, c, 3, 3)
    for i in (1, 2, 3, 4):
        p = f"scratch.refinenet{i}."
        sp[p + "out_conv.weight"] = (FEATURES, FEATURES, 1, 1)
        sp[p + "out_conv.bias"] = (FEATURES,)
        for u in (1, 2):
            if u == 1 and i == 4 and not include_unused:
                continue
            for c in (1, 2):
                sp[f"{p}resConfUnit{u}.conv{c}.weight"] = (FEATURES, FEATURES, 3, 3)
                sp[f"{p}resConfUnit{u}.conv{c}.bias"] = (FEATURES,)
    oc = "scratch.output_conv."
    sp[oc + "0.weight"] = (FEATURES // 2, FEATURES, 3, 3)
    sp[oc + "0.bias"] = (FEATURES // 2,)
    sp[oc + "2.weight"] = (32, FEATURES // 2, 3, 3)
    sp[oc + "2.bias"] = (32,)
    sp[oc + "4.weight"] = (num_channels, 32, 1, 1)
    sp[oc + "4.bias"] = (num_channels,)
    return sp


UNUSED_KEYS = (
    "pretrained.model.norm.", "pretrained.model.head.", "scratch.refinenet4.resConfUnit1.",
    "depth.scratch.refinenet4.resConfUnit1.",
)


def is_unused(key: str) -> bool:
    return key.startswith(UNUSED_KEYS)


FAMILIES = ("default", "trained")


def _gain_for(key: str, shape, family: str = "default") -> Tuple[str, float]:
    """Returns (kind, scale) for the synthetic init of one tensor.

    family 'default': every residual branch carries about as much signal as its shortcut (norm3 gamma 0.5, proj / fc2
    gain 0.5) -- a chaotic random residual net that amplifies a perturbation ~60x through ResNet stage 2.
    family 'trained': the conditioning of a trained network -- the last norm of every bottleneck has a small gamma
    (timm zero-inits it; trained values stay small) and the ViT branch outputs are scaled by 1/sqrt(2*depth) -- so
    that residual branches are corrections to the stream, as in the published checkpoints."""
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    if key.endswith("cls_token"):
        return "normal", 0.5
    if key.endswith("pos_embed"):
        return "normal", 0.5
    if ".norm" in key or key.endswith("norm.weight") or key.endswith("norm.bias"):
        # GroupNorm / LayerNorm affine
        if key.endswith(".weight"):
            if "norm3" in key:
                return ("gamma", 0.2 if family == "trained" else 0.5)
            return ("gamma", 1.0)
        return "normal", 0.1
    if key.endswith(".bias"):
        if key == "scratch.output_conv.4.bias":
            return "const", 0.5
        return "normal", 0.1
    # weights
    if "patch_embed.backbone" in key:
        return "normal", 0.05  # standardised at run time: scale-free
    if "act_postprocess" in key and key.endswith(".4.weight") and len(shape) == 4 and shape[2] in (2, 4):
        return "normal", 1.0 / (shape[0] ** 0.5)  # ConvTranspose2d [Cin, Cout, k, k], k == stride: one tap per output pixel
    g = 1.0
    if "attn.proj" in key or "mlp.fc2" in key:
        g = (1.0 / (2 * VIT_DEPTH) ** 0.5) if family == "trained" else 0.5
    elif "resConfUnit" in key:
        g = 1.0
    elif "project.0" in key:
        g = 0.5
    elif key == "scratch.output_conv.4.weight":
        g = 0.12
    elif key in ("scratch.output_conv.0.weight", "scratch.output_conv.2.weight"):
        g = 1.4
    return "normal", g / (fan_in ** 0.5)


def random_state_dict(seed: int = 0, num_channels: int = 3, include_unused: bool = True,
                      family: str = "default", backbone: str = "vitb_rn50_384") -> Dict[str, torch.Tensor]:
    """Deterministic synthetic fp32 weights (CPU generator => identical on every host).  `family`: see _gain_for."""
    if family not in FAMILIES:
        raise ValueError(f"family must be one of {FAMILIES}")
    g = torch.Generator(device="cpu")
    g.manual_seed(1000003 * int(seed) + 17)
    sd: Dict[str, torch.Tensor] = OrderedDict()
    for key, shape in state_dict_spec(num_channels, include_unused, backbone).items():
        kind, s = _gain_for(key, shape, family)
        if backbone == "vitl16_384" and ("attn.proj" in key or "mlp.fc2" in key) and key.endswith(".weight") and family == "default":
            s *= 0.7  # 24 residual blocks instead of 12: keep the token stream O(1)
        if kind == "const":
            t = torch.full(shape, s, dtype=torch.float32)
        elif kind == "gamma":
            t = s * (1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float32))
        else:
            t = s * torch.randn(shape, generator=g, dtype=torch.float32)
        sd[key] = t
    return sd


DUAL_PREFIX = "depth."  # second decoder of the dual-task engine: depth.scratch.* (include/dptx.h dptx_forward_dual)


def dual_state_dict_spec(include_unused: bool = True) -> "OrderedDict[str, Tuple[int, ...]]":
    """Keys of the dual-task engine: the normal model's full spec (shared ``pretrained.*`` + ``scratch.*`` with 3
    channels) followed by the depth decoder's ``scratch.*`` (1 channel) behind the ``depth.`` prefix."""
    sp = state_dict_spec(3, include_unused)
    for k, shp in state_dict_spec(1, include_unused).items():
        if k.startswith("scratch."):
            sp[DUAL_PREFIX + k] = shp
    return sp


def compose_dual_state_dict(normal_sd: Dict[str, torch.Tensor], depth_sd: Dict[str, torch.Tensor],
                            backbone: str = "normal") -> Dict[str, torch.Tensor]:
    """Builds the dual-task state dict from two single-task ones: ``pretrained.*`` from ``backbone`` ('normal' or
    'depth'), ``scratch.*`` from the normal model, ``depth.scratch.*`` from the depth model."""
    if backbone not in ("normal", "depth"):
        raise ValueError("backbone must be 'normal' or 'depth'")
    src = normal_sd if backbone == "normal" else depth_sd
    sd: Dict[str, torch.Tensor] = OrderedDict((k, v) for k, v in src.items() if k.startswith("pretrained."))
    sd.update((k, v) for k, v in normal_sd.items() if k.startswith("scratch."))
    sd.update((DUAL_PREFIX + k, v) for k, v in depth_sd.items() if k.startswith("scratch."))
    return sd


def split_dual_state_dict(sd: Dict[str, torch.Tensor]):
    """Inverse view used by the oracle: (normal single-task sd, depth single-task sd) sharing ``pretrained.*``."""
    shared = OrderedDict((k, v) for k, v in sd.items() if k.startswith("pretrained."))
    normal = OrderedDict(shared)
    normal.update((k, v) for k, v in sd.items() if k.startswith("scratch."))
    depth = OrderedDict(shared)
    depth.update((k[len(DUAL_PREFIX):], v) for k, v in sd.items() if k.startswith(DUAL_PREFIX))
    return normal, depth


def random_dual_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded dual-task weights: encoder + normal decoder of ``random_state_dict(seed, 3)``, depth decoder of
    ``random_state_dict(seed + 1000, 1)``."""
    return compose_dual_state_dict(random_state_dict(seed, 3), random_state_dict(seed + 1000, 1))


def read_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """Reads a reference checkpoint into a plain {key: fp32 tensor} dict.

    Accepts what the reference accepts: a Lightning ``.ckpt`` whose ``state_dict`` keys
    carry a 6-char ``model.`` prefix (demo.py:65-68), a raw state_dict (demo.py:69-70),
    or MiDaS' ``{"optimizer":..., "model":...}`` wrapper (base_model.py:11-16).
    """
    obj = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(obj, dict) and "state_dict" in obj:
        sd = OrderedDict((k[6:], v) for k, v in obj["state_dict"].items())
    elif isinstance(obj, dict) and "optimizer" in obj and "model" in obj:
        sd = obj["model"]
    else:
        sd = obj
    return OrderedDict((k, v.detach().float().contiguous()) for k, v in sd.items())


def synthetic_input(seed: int, batch: int, task: str = "normal", size=384) -> torch.Tensor:
    """Seeded synthetic image batch in the reference's input convention:
    normal: [0,1] (demo.py:74-76); depth: [-1,1] (demo.py:92-95 Normalize(0.5,0.5))."""
    g = torch.Generator(device="cpu")
    g.manual_seed(7919 * int(seed) + 11)
    h, w = (size, size) if isinstance(size, int) else size
    x = torch.rand(batch, 3, h, w, generator=g, dtype=torch.float32)
    if task == "depth":
        x = 2.0 * x - 1.0
    return x
