"""Stand-in for ``timm==0.4.12`` exposing ONLY ``create_model('vit_base_resnet50_384')``.

TEST INFRASTRUCTURE ONLY (see oracle/dpt_oracle.py header).  The reference imports
``timm`` at ``omnidata_tools/torch/modules/midas/vit.py:3`` and calls
``timm.create_model("vit_base_resnet50_384", pretrained=...)`` at ``vit.py:483``; timm is
pinned in ``requirements.txt:15`` but is neither vendored nor installable here.  This shim
restates that one model as ``nn.Module``s with timm's attribute and state_dict names
(``patch_embed.backbone.stem.conv`` ..., ``blocks.{l}.attn.qkv`` ...)
so that the reference's *own, unmodified* ``dpt_depth.py / blocks.py / vit.py`` can run in
this container.  It is written module-style, independently of the functional restatement
in ``dpt_oracle.py``; ``validate_vs_reference.py`` checks the two against each other.

Install with ``install()`` -> ``sys.modules['timm']``.
"""
from __future__ import annotations

import math
import sys
import types
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F


def _same_pad(i, k, s, d=1):
    return max((math.ceil(i / s) - 1) * s + (k - 1) * d + 1 - i, 0)


def _pad_same(x, k, s, value=0.0):
    ih, iw = x.shape[-2:]
    ph, pw = _same_pad(ih, k[0], s[0]), _same_pad(iw, k[1], s[1])
    if ph > 0 or pw > 0:
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2], value=value)
    return x


class StdConv2dSame(nn.Conv2d):
    """timm 0.4.x layers/std_conv.py: weight-standardised conv with TF-SAME padding."""

    ws_form = "timm04"  # class-level switch used by the HF cross-check

    def __init__(self, cin, cout, kernel_size, stride=1, eps=1e-5):
        super().__init__(cin, cout, kernel_size, stride=stride, padding=0, bias=False)
        self.eps = eps

    def get_weight(self):
        std, mean = torch.std_mean(self.weight, dim=[1, 2, 3], keepdim=True, unbiased=False)
        if self.ws_form == "timm04":
            return (self.weight - mean) / (std + self.eps)
        return (self.weight - mean) / torch.sqrt(std * std + self.eps)

    def forward(self, x):
        x = _pad_same(x, self.kernel_size, self.stride)
        return F.conv2d(x, self.get_weight(), None, self.stride, (0, 0))


class GroupNormAct(nn.GroupNorm):
    def __init__(self, num_channels, num_groups=32, eps=1e-5, apply_act=True):
        super().__init__(num_groups, num_channels, eps=eps)
        self.apply_act = apply_act

    def forward(self, x):
        x = F.group_norm(x, self.num_groups, self.weight, self.bias, self.eps)
        return F.relu(x) if self.apply_act else x


class MaxPool2dSame(nn.MaxPool2d):
    def forward(self, x):
        x = _pad_same(x, (self.kernel_size,) * 2, (self.stride,) * 2, value=-float("inf"))
        return F.max_pool2d(x, self.kernel_size, self.stride, (0, 0))


class DownsampleConv(nn.Module):
    def __init__(self, cin, cout, stride, conv_layer):
        super().__init__()
        self.conv = conv_layer(cin, cout, 1, stride=stride)
        self.norm = GroupNormAct(cout, apply_act=False)

    def forward(self, x):
        return self.norm(self.conv(x))


class Bottleneck(nn.Module):
    """Non pre-activation bottleneck (timm resnetv2.Bottleneck, 'Used for ViT')."""

    def __init__(self, cin, cout, stride, conv_layer, proj):
        super().__init__()
        mid = cout // 4
        self.downsample = DownsampleConv(cin, cout, stride, conv_layer) if proj else None
        self.conv1 = conv_layer(cin, mid, 1)
        self.norm1 = GroupNormAct(mid)
        self.conv2 = conv_layer(mid, mid, 3, stride=stride)
        self.norm2 = GroupNormAct(mid)
        self.conv3 = conv_layer(mid, cout, 1)
        self.norm3 = GroupNormAct(cout, apply_act=False)

    def forward(self, x):
        shortcut = x if self.downsample is None else self.downsample(x)
        x = self.norm1(self.conv1(x))
        x = self.norm2(self.conv2(x))
        x = self.norm3(self.conv3(x))
        return F.relu(x + shortcut)


class ResNetStage(nn.Module):
    def __init__(self, cin, cout, stride, depth, conv_layer):
        super().__init__()
        blocks = []
        for b in range(depth):
            blocks.append(Bottleneck(cin, cout, stride if b == 0 else 1, conv_layer, proj=(b == 0)))
            cin = cout
        self.blocks = nn.Sequential(*blocks)

    def forward(self, x):
        return self.blocks(x)


class ResNetV2(nn.Module):
    """layers=(3,4,9), preact=False, stem_type='same', num_classes=0, global_pool=''."""

    def __init__(self, layers=(3, 4, 9), conv_layer=None):
        super().__init__()
        self.stem = nn.Sequential()
        self.stem.add_module("conv", conv_layer(3, 64, 7, stride=2))
        self.stem.add_module("norm", GroupNormAct(64))
        self.stem.add_module("pool", MaxPool2dSame(3, 2))
        stages, cin = [], 64
        for cout, stride, depth in zip((256, 512, 1024), (1, 2, 2), layers):
            stages.append(ResNetStage(cin, cout, stride, depth, conv_layer))
            cin = cout
        self.stages = nn.Sequential(*stages)
        self.norm = nn.Identity()

    def forward(self, x):
        return self.norm(self.stages(self.stem(x)))


class HybridEmbed(nn.Module):
    def __init__(self, backbone, embed_dim=768):
        super().__init__()
        self.backbone = backbone
        self.proj = nn.Conv2d(1024, embed_dim, kernel_size=1, stride=1)

    def forward(self, x):
        x = self.backbone(x)
        return self.proj(x).flatten(2).transpose(1, 2)


class Attention(nn.Module):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        attn = (q @ k.transpose(-2, -1)) * self.scale
        attn = attn.softmax(dim=-1)
        return self.proj((attn @ v).transpose(1, 2).reshape(B, N, C))


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class VisionTransformer(nn.Module):
    def __init__(self):
        super().__init__()
        conv_layer = partial(StdConv2dSame, eps=1e-8)
        self.patch_embed = HybridEmbed(ResNetV2((3, 4, 9), conv_layer))
        self.cls_token = nn.Parameter(torch.zeros(1, 1, 768))
        self.pos_embed = nn.Parameter(torch.zeros(1, 577, 768))
        self.pos_drop = nn.Dropout(p=0.0)
        self.blocks = nn.ModuleList([Block(768, 12) for _ in range(12)])
        self.norm = nn.LayerNorm(768, eps=1e-6)
        self.head = nn.Linear(768, 1000)  # unused by DPT, but present in checkpoints


class PatchEmbed(nn.Module):
    """timm 0.4.12 PatchEmbed: Conv2d(3, embed_dim, kernel_size=patch, stride=patch) -> flatten(2).transpose(1, 2)."""

    def __init__(self, patch_size=16, embed_dim=1024):
        super().__init__()
        self.proj = nn.Conv2d(3, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        return self.proj(x).flatten(2).transpose(1, 2)


class PlainVisionTransformer(nn.Module):
    """timm 0.4.12 vit_large_patch16_384: patch 16, embed 1024, depth 24, heads 16, mlp_ratio 4, qkv_bias, LayerNorm
    eps 1e-6 (the reference consumes .patch_embed.proj, .cls_token, .pos_embed, .pos_drop, .blocks, .norm: vit.py:119-155)."""

    def __init__(self, embed_dim=1024, depth=24, num_heads=16):
        super().__init__()
        self.patch_embed = PatchEmbed(16, embed_dim)
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, 577, embed_dim))
        self.pos_drop = nn.Dropout(p=0.0)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.head = nn.Linear(embed_dim, 1000)  # unused by DPT, but present in checkpoints


def create_model(name, pretrained=False, **kwargs):
    # `pretrained` is ignored: dpt_depth.py:51 hard-codes True, which would download.
    if name == "vit_base_resnet50_384":
        return VisionTransformer()
    if name == "vit_large_patch16_384":
        return PlainVisionTransformer(1024, 24, 16)
    raise RuntimeError(f"timm shim provides vit_base_resnet50_384 and vit_large_patch16_384, not {name}")


def install():
    mod = types.ModuleType("timm")
    mod.create_model = create_model
    mod.__version__ = "0.4.12-shim"
    sys.modules["timm"] = mod
    return mod
