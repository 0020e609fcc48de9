"""CPU fp32 ORACLE for the DPT-Hybrid-384 forward pass.  TEST INFRASTRUCTURE ONLY.

This file is a restatement, in plain functional PyTorch fp32, of the one hot path
named in BASELINE.json: ``DPTDepthModel(backbone='vitb_rn50_384').forward``.  It is
the checker used by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` -- it is never imported by the product package ``omnidata_amd``.

What it follows (paths relative to /root/reference/omnidata_tools/torch):

* ``modules/midas/dpt_depth.py:67-85``   DPT.forward wiring (encoder -> layerN_rn ->
  refinenet4..1 -> output_conv) and ``:87-107`` the task head + ``squeeze(dim=1)``.
* ``modules/midas/blocks.py:231-288``    ResidualConvUnit_custom (bn=False),
  ``:291-341`` FeatureFusionBlock_custom (deconv=False, expand=False,
  align_corners=True), ``:49-75`` _make_scratch (3x3, pad 1, no bias).
* ``modules/midas/vit.py:61-99``         forward_vit, ``:119-155`` forward_flex,
  ``:36-47`` ProjectReadout, ``:431-462`` act_postprocess3/4.
* timm==0.4.12 ``vit_base_resnet50_384`` (requirements.txt:15, call site vit.py:483).
  timm is NOT vendored in the reference and NOT installable here, so its published
  algorithm is restated below (ResNetV2 layers (3,4,9), StdConv2dSame eps=1e-8,
  GroupNormAct(32, eps=1e-5), MaxPool2dSame, HybridEmbed 1x1 proj, 12x ViT Block with
  LayerNorm eps=1e-6, erf-GELU).  That part is **parity unpinned** against real timm;
  it is cross-validated against the independent HuggingFace ``BitBackbone`` /
  ``DPTForDepthEstimation`` implementation in ``oracle/validate_vs_hf.py`` and the
  reference-side wiring is pinned exactly against the reference's own modules in
  ``oracle/validate_vs_reference.py`` (which also emits ``tests/golden/*.npz``).

The state_dict key names are the reference's, after stripping the
Lightning ``model.`` prefix (demo.py:65-70).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def oracle_threads(cap: int = 32) -> int:
    """Threads for the CPU oracle: oneDNN/MKL convs at batch 1-4 get SLOWER past a few dozen
    threads (a 256-thread run on the 2x64-core GPU host took 28 s/image), so cap them."""
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    n = max(1, min(n, cap))
    torch.set_num_threads(n)
    return n

# ResNetV2 hybrid backbone geometry (timm vit_base_r50_s16_384: layers=(3,4,9))
STAGE_DEPTHS = (3, 4, 9)
STAGE_OUT = (256, 512, 1024)
STAGE_STRIDE = (1, 2, 2)
VIT_DEPTH = 12
VIT_DIM = 768
VIT_HEADS = 12
HOOK_BLOCKS = (8, 11)  # dpt_depth.py:41-45 hooks [0,1,8,11]: "3","4" sit on blocks 8, 11


# --------------------------------------------------------------------------- timm bits
def same_pad_amount(i: int, k: int, s: int, d: int = 1) -> int:
    """TF 'SAME' total padding (timm padding.py get_same_padding)."""
    return max((math.ceil(i / s) - 1) * s + (k - 1) * d + 1 - i, 0)


def pad_same(x: Tensor, k: int, s: int, value: float = 0.0) -> Tensor:
    ih, iw = x.shape[-2:]
    ph, pw = same_pad_amount(ih, k, s), same_pad_amount(iw, k, s)
    if ph > 0 or pw > 0:
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2], value=value)
    return x


def standardize_weight(w: Tensor, eps: float = 1e-8, form: str = "timm04") -> Tensor:
    """Weight standardisation of StdConv2dSame.

    form='timm04': (w-mean)/(std+eps)        -- timm 0.4.x std_conv.py get_weight
    form='hf'    : (w-mean)/sqrt(var+eps)    -- later timm / HF modeling_bit.py:118-125
    Biased variance over (Cin,kh,kw) per output channel in both.
    """
    std, mean = torch.std_mean(w, dim=[1, 2, 3], keepdim=True, unbiased=False)
    if form == "timm04":
        return (w - mean) / (std + eps)
    if form == "hf":
        return (w - mean) / torch.sqrt(std * std + eps)
    raise ValueError(form)


def std_conv_same(x: Tensor, w: Tensor, stride: int, ws_eps: float, ws_form: str) -> Tensor:
    k = w.shape[-1]
    x = pad_same(x, k, stride)
    return F.conv2d(x, standardize_weight(w, ws_eps, ws_form), None, stride)


def gn(x: Tensor, sd: Dict[str, Tensor], key: str, relu: bool) -> Tensor:
    y = F.group_norm(x, 32, sd[key + ".weight"], sd[key + ".bias"], 1e-5)
    return F.relu(y) if relu else y


def resnetv2_backbone(x: Tensor, sd: Dict[str, Tensor], pre: str, taps: Optional[dict],
                      ws_eps: float, ws_form: str):
    """timm ResNetV2(layers=(3,4,9), preact=False, stem_type='same', StdConv2dSame).

    Returns (stage0_out, stage1_out, stage2_out); the first two are the forward-hook
    activations "1" and "2" (vit.py:363-368).
    """
    conv = lambda t, key, s: std_conv_same(t, sd[pre + key + ".weight"], s, ws_eps, ws_form)
    # stem: conv7x7 s2 SAME -> GN+ReLU -> MaxPool2dSame(3, 2) (pad value -inf)
    x = conv(x, "stem.conv", 2)
    x = gn(x, sd, pre + "stem.norm", True)
    x = F.max_pool2d(pad_same(x, 3, 2, value=-float("inf")), 3, 2)
    if taps is not None:
        taps["stem"] = x
    outs = []
    for s, depth in enumerate(STAGE_DEPTHS):
        for b in range(depth):
            bp = f"stages.{s}.blocks.{b}."
            stride = STAGE_STRIDE[s] if b == 0 else 1
            shortcut = x
            if b == 0:  # DownsampleConv: 1x1 StdConv(stride) + GN (no act)
                shortcut = gn(conv(x, bp + "downsample.conv", stride), sd, pre + bp + "downsample.norm", False)
            y = gn(conv(x, bp + "conv1", 1), sd, pre + bp + "norm1", True)
            y = gn(conv(y, bp + "conv2", stride), sd, pre + bp + "norm2", True)  # V1.5: stride on conv2
            y = gn(conv(y, bp + "conv3", 1), sd, pre + bp + "norm3", False)
            x = F.relu(y + shortcut)
        outs.append(x)
        if taps is not None:
            taps[f"s{s}"] = x
    return outs


def vit_block(x: Tensor, sd: Dict[str, Tensor], p: str, heads: int = VIT_HEADS) -> Tensor:
    """timm Block: x += attn(norm1(x)); x += mlp(norm2(x)); LayerNorm eps 1e-6."""
    B, N, C = x.shape
    h = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, heads, C // heads).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * ((C // heads) ** -0.5)
    attn = attn.softmax(dim=-1)
    h = (attn @ v).transpose(1, 2).reshape(B, N, C)
    x = x + F.linear(h, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
    h = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    h = F.gelu(F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))  # exact erf GELU
    return x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])


# ------------------------------------------------------------------ reference-side bits
def project_readout(tok: Tensor, sd: Dict[str, Tensor], p: str) -> Tensor:
    """vit.py:36-47 ProjectReadout(start_index=1): GELU(Linear(cat(tok[1:], cls)))."""
    readout = tok[:, 0].unsqueeze(1).expand_as(tok[:, 1:])
    feats = torch.cat((tok[:, 1:], readout), -1)
    return F.gelu(F.linear(feats, sd[p + "project.0.weight"], sd[p + "project.0.bias"]))


def rcu(x: Tensor, sd: Dict[str, Tensor], p: str) -> Tensor:
    """blocks.py:263-286 ResidualConvUnit_custom, bn=False, activation=ReLU(False)."""
    out = F.relu(x)
    out = F.conv2d(out, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    out = F.relu(out)
    out = F.conv2d(out, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    return out + x


def fusion(sd: Dict[str, Tensor], p: str, x0: Tensor, x1: Optional[Tensor] = None) -> Tensor:
    """blocks.py:320-341 FeatureFusionBlock_custom.forward."""
    out = x0
    if x1 is not None:
        out = out + rcu(x1, sd, p + "resConfUnit1.")
    out = rcu(out, sd, p + "resConfUnit2.")
    out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    return F.conv2d(out, sd[p + "out_conv.weight"], sd[p + "out_conv.bias"])


@torch.no_grad()
def dpt_forward(sd: Dict[str, Tensor], x: Tensor, taps: Optional[dict] = None,
                ws_eps: float = 1e-8, ws_form: str = "timm04",
                non_negative: bool = True) -> Tensor:
    """Full forward.  x: [B,3,H,W] fp32 (H,W multiples of 16; 384 on the benchmark).

    Returns [B,3,H,W] for the normal head (num_channels=3) or [B,H,W] for depth
    (num_channels=1; dpt_depth.py:106-107 squeeze(dim=1)).  ``taps`` (optional dict)
    receives the stage activations named in SURVEY.md A.1, all in NCHW / [B,N,C] fp32.
    """
    x = x.float()
    B, _, H, W = x.shape
    gh, gw = H // 16, W // 16
    vp = "pretrained.model."
    # --- forward_flex (vit.py:119-155)
    pos = sd[vp + "pos_embed"]
    g_old = int(math.sqrt(pos.shape[1] - 1))
    if (gh, gw) != (g_old, g_old):  # _resize_pos_embed vit.py:102-116 (identity at 384)
        grid = pos[0, 1:].reshape(1, g_old, g_old, -1).permute(0, 3, 1, 2)
        grid = F.interpolate(grid, size=(gh, gw), mode="bilinear")
        pos = torch.cat([pos[:, :1], grid.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)], dim=1)
    s0, s1, s2 = resnetv2_backbone(x, sd, vp + "patch_embed.backbone.", taps, ws_eps, ws_form)
    t = F.conv2d(s2, sd[vp + "patch_embed.proj.weight"], sd[vp + "patch_embed.proj.bias"])
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat((sd[vp + "cls_token"].expand(B, -1, -1), t), dim=1) + pos
    if taps is not None:
        taps["tok0"] = t
    hooks = {}
    for l in range(VIT_DEPTH):
        t = vit_block(t, sd, f"{vp}blocks.{l}.")
        if taps is not None:
            taps[f"blk{l}"] = t
        if l in HOOK_BLOCKS:
            hooks[l] = t
    # final model.norm output is discarded by forward_vit (vit.py:64,153): dead compute.
    # --- forward_vit post-processing (vit.py:66-97); act_postprocess1/2 are Identity
    layer_1, layer_2 = s0, s1
    pp = "pretrained.act_postprocess"
    l3 = project_readout(hooks[HOOK_BLOCKS[0]], sd, pp + "3.0.").transpose(1, 2).reshape(B, VIT_DIM, gh, gw)
    layer_3 = F.conv2d(l3, sd[pp + "3.3.weight"], sd[pp + "3.3.bias"])
    l4 = project_readout(hooks[HOOK_BLOCKS[1]], sd, pp + "4.0.").transpose(1, 2).reshape(B, VIT_DIM, gh, gw)
    l4 = F.conv2d(l4, sd[pp + "4.3.weight"], sd[pp + "4.3.bias"])
    layer_4 = F.conv2d(l4, sd[pp + "4.4.weight"], sd[pp + "4.4.bias"], stride=2, padding=1)
    # --- DPT.forward (dpt_depth.py:73-83)
    l1rn = F.conv2d(layer_1, sd["scratch.layer1_rn.weight"], None, padding=1)
    l2rn = F.conv2d(layer_2, sd["scratch.layer2_rn.weight"], None, padding=1)
    l3rn = F.conv2d(layer_3, sd["scratch.layer3_rn.weight"], None, padding=1)
    l4rn = F.conv2d(layer_4, sd["scratch.layer4_rn.weight"], None, padding=1)
    p4 = fusion(sd, "scratch.refinenet4.", l4rn)  # resConfUnit1 of refinenet4 is never used
    p3 = fusion(sd, "scratch.refinenet3.", p4, l3rn)
    p2 = fusion(sd, "scratch.refinenet2.", p3, l2rn)
    p1 = fusion(sd, "scratch.refinenet1.", p2, l1rn)
    # --- head (dpt_depth.py:91-99)
    oc = "scratch.output_conv."
    h0 = F.conv2d(p1, sd[oc + "0.weight"], sd[oc + "0.bias"], padding=1)
    h0u = F.interpolate(h0, scale_factor=2, mode="bilinear", align_corners=True)
    h1 = F.relu(F.conv2d(h0u, sd[oc + "2.weight"], sd[oc + "2.bias"], padding=1))
    pre = F.conv2d(h1, sd[oc + "4.weight"], sd[oc + "4.bias"])
    out = F.relu(pre) if non_negative else pre
    if taps is not None:
        taps.update(l3=layer_3, l4=layer_4, l1_rn=l1rn, l2_rn=l2rn, l3_rn=l3rn, l4_rn=l4rn,
                    p4=p4, p3=p3, p2=p2, p1=p1, h0=h0, h1=h1, pre=pre, out=out)
    return out.squeeze(dim=1)


# ----------------------------------------------------------------- DPT-Large (vitl16_384)
VITL_HOOKS = (5, 11, 17, 23)  # dpt_depth.py:41-45


def _decoder_and_head(sd: Dict[str, Tensor], layers, taps: Optional[dict], non_negative: bool) -> Tensor:
    """DPT.forward after forward_vit (dpt_depth.py:73-83) + the head (dpt_depth.py:91-99); shared with the hybrid."""
    l1rn = F.conv2d(layers[0], sd["scratch.layer1_rn.weight"], None, padding=1)
    l2rn = F.conv2d(layers[1], sd["scratch.layer2_rn.weight"], None, padding=1)
    l3rn = F.conv2d(layers[2], sd["scratch.layer3_rn.weight"], None, padding=1)
    l4rn = F.conv2d(layers[3], sd["scratch.layer4_rn.weight"], None, padding=1)
    p4 = fusion(sd, "scratch.refinenet4.", l4rn)
    p3 = fusion(sd, "scratch.refinenet3.", p4, l3rn)
    p2 = fusion(sd, "scratch.refinenet2.", p3, l2rn)
    p1 = fusion(sd, "scratch.refinenet1.", p2, l1rn)
    oc = "scratch.output_conv."
    h0 = F.conv2d(p1, sd[oc + "0.weight"], sd[oc + "0.bias"], padding=1)
    h0u = F.interpolate(h0, scale_factor=2, mode="bilinear", align_corners=True)
    h1 = F.relu(F.conv2d(h0u, sd[oc + "2.weight"], sd[oc + "2.bias"], padding=1))
    pre = F.conv2d(h1, sd[oc + "4.weight"], sd[oc + "4.bias"])
    out = F.relu(pre) if non_negative else pre
    if taps is not None:
        taps.update(l1=layers[0], l2=layers[1], l3=layers[2], l4=layers[3], l1_rn=l1rn, l2_rn=l2rn, l3_rn=l3rn, l4_rn=l4rn,
                    p4=p4, p3=p3, p2=p2, p1=p1, h0=h0, h1=h1, pre=pre, out=out)
    return out.squeeze(dim=1)


@torch.no_grad()
def dpt_forward_vitl16(sd: Dict[str, Tensor], x: Tensor, taps: Optional[dict] = None, non_negative: bool = True) -> Tensor:
    """``DPTDepthModel(backbone='vitl16_384')`` (DPT-Large; demo.py:81, blocks.py:12-18): forward_flex on timm's
    ``vit_large_patch16_384`` (vit.py:119-155: 16x16 stride-16 patch conv, cls, resized pos_embed, 24 blocks of width 1024
    with 16 heads), hooks on blocks 5/11/17/23 (vit.py:299-309, dpt_depth.py:41-45), ProjectReadout + reassemble
    (vit.py:176-260: 1x1 conv then ConvTranspose2d k4s4 / k2s2 / identity / Conv2d 3x3 s2 p1), then the decoder and head
    the hybrid model uses.  State-dict keys: omnidata_amd.weights.vitl16_state_dict_spec."""
    x = x.float()
    B, _, H, W = x.shape
    gh, gw = H // 16, W // 16
    D, heads = 1024, 16
    vp = "pretrained.model."
    pos = sd[vp + "pos_embed"]
    g_old = int(math.sqrt(pos.shape[1] - 1))
    if (gh, gw) != (g_old, g_old):  # _resize_pos_embed vit.py:102-116
        grid = pos[0, 1:].reshape(1, g_old, g_old, -1).permute(0, 3, 1, 2)
        grid = F.interpolate(grid, size=(gh, gw), mode="bilinear")
        pos = torch.cat([pos[:, :1], grid.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)], dim=1)
    t = F.conv2d(x, sd[vp + "patch_embed.proj.weight"], sd[vp + "patch_embed.proj.bias"], stride=16)
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat((sd[vp + "cls_token"].expand(B, -1, -1), t), dim=1) + pos
    if taps is not None:
        taps["tok0"] = t
    hooks = {}
    for l in range(24):
        t = vit_block(t, sd, f"{vp}blocks.{l}.", heads)
        if taps is not None:
            taps[f"blk{l}"] = t
        if l in VITL_HOOKS:
            hooks[l] = t
    pp = "pretrained.act_postprocess"
    layers = []
    for n, l in enumerate(VITL_HOOKS, start=1):
        r = project_readout(hooks[l], sd, f"{pp}{n}.0.").transpose(1, 2).reshape(B, D, gh, gw)
        y = F.conv2d(r, sd[f"{pp}{n}.3.weight"], sd[f"{pp}{n}.3.bias"])
        if n == 1:
            y = F.conv_transpose2d(y, sd[f"{pp}1.4.weight"], sd[f"{pp}1.4.bias"], stride=4)
        elif n == 2:
            y = F.conv_transpose2d(y, sd[f"{pp}2.4.weight"], sd[f"{pp}2.4.bias"], stride=2)
        elif n == 4:
            y = F.conv2d(y, sd[f"{pp}4.4.weight"], sd[f"{pp}4.4.bias"], stride=2, padding=1)
        layers.append(y)
    return _decoder_and_head(sd, layers, taps, non_negative)


# ------------------------------------------------------- parity metrics
def ssi_align(pred: Tensor, target: Tensor) -> Tensor:
    """Least-squares scale+shift alignment of pred to target per image
    (losses/midas_loss.py:10-30 compute_scale_and_shift, mask = all ones)."""
    p = pred.reshape(pred.shape[0], -1).double()
    t = target.reshape(target.shape[0], -1).double()
    n = p.shape[1]
    a00, a01, a11 = (p * p).sum(1), p.sum(1), torch.full_like(p[:, 0], n)
    b0, b1 = (p * t).sum(1), t.sum(1)
    det = a00 * a11 - a01 * a01
    scale = torch.where(det > 0, (a11 * b0 - a01 * b1) / det, torch.ones_like(det))
    shift = torch.where(det > 0, (-a01 * b0 + a00 * b1) / det, torch.zeros_like(det))
    return (scale[:, None] * p + shift[:, None]).reshape(pred.shape).float()


def mean_angular_error_deg(pred: Tensor, target: Tensor) -> float:
    """Mean angle between normal vectors encoded as rgb in [0,1] ([B,3,H,W]);
    follows paper_code/evaluation_metrics.py:34-45 (vectors = 2*rgb-1)."""
    a = F.normalize(pred.double() * 2 - 1, dim=1, eps=1e-12)
    b = F.normalize(target.double() * 2 - 1, dim=1, eps=1e-12)
    cos = (a * b).sum(1).clamp(-1, 1)
    return float(torch.rad2deg(torch.acos(cos)).mean())


def dpt_forward_dual(sd, x, taps_normal=None, taps_depth=None, **kw):
    """Dual-task oracle: the reference forward (dpt_depth.py:67-85) run twice on the same
    input, once per decoder, with `pretrained.*` tied -- which is what a shared-encoder engine must reproduce.
    `sd` uses the dual key layout of omnidata_amd.weights.dual_state_dict_spec.  Returns (normal, depth)."""
    from omnidata_amd.weights import split_dual_state_dict
    normal_sd, depth_sd = split_dual_state_dict(sd)
    return dpt_forward(normal_sd, x, taps_normal, **kw), dpt_forward(depth_sd, x, taps_depth, **kw)
