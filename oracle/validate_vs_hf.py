"""Cross-validates the restated timm backbone in oracle/dpt_oracle.py against an
INDEPENDENT implementation: HuggingFace ``DPTForDepthEstimation(is_hybrid=True)`` with
its ``BitBackbone`` (transformers/models/dpt/modeling_dpt.py, models/bit/modeling_bit.py),
which was converted from the same DPT-Hybrid family.  TEST INFRASTRUCTURE ONLY.

The same seeded weights (omnidata_amd.weights.random_state_dict, depth head C=1) are
key-mapped into the HF model and both are run on the same input.  Agreement validates the
parts the reference delegates to timm==0.4.12 and therefore cannot be pinned against the
reference itself: TF-SAME asymmetric padding, stride-on-conv2 bottlenecks, GroupNorm(32)
eps 1e-5, max-pool padding, qkv packing order, LayerNorm eps 1e-6, erf-GELU, readout and
fusion order.  Known deliberate difference: HF standardises weights as
(w-mean)/sqrt(var+1e-8) whereas timm 0.4.x uses (w-mean)/(std+1e-8); the oracle is run
with both forms and the gap between the forms is reported (it is ~1e-6).

Usage: python -m oracle.validate_vs_hf
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.dpt_oracle import dpt_forward  # noqa: E402
from omnidata_amd.weights import random_state_dict, synthetic_input  # noqa: E402


def to_hf(sd, hybrid=True):
    out = {}
    vp = "pretrained.model."
    for k, v in sd.items():
        if k.startswith(vp + "head."):
            continue
        if k == vp + "cls_token":
            out["dpt.embeddings.cls_token"] = v
        elif k == vp + "pos_embed":
            out["dpt.embeddings.position_embeddings"] = v
        elif k.startswith(vp + "patch_embed.backbone."):
            r = k[len(vp + "patch_embed.backbone."):]
            r = r.replace("stem.conv.", "embedder.convolution.").replace("stem.norm.", "embedder.norm.")
            if r.startswith("stages."):
                r = "encoder." + r.replace(".blocks.", ".layers.")
            out["dpt.embeddings.backbone.bit." + r] = v
        elif k.startswith(vp + "patch_embed.proj."):
            # hybrid: 1x1 projection of the ResNet map; DPT-Large: timm PatchEmbed = HF DPTViTPatchEmbeddings
            out[("dpt.embeddings.projection." if hybrid else "dpt.embeddings.patch_embeddings.projection.") + k.rsplit(".", 1)[1]] = v
        elif k.startswith(vp + "blocks."):
            _, _, _, l, rest = k.split(".", 4)
            p = f"dpt.encoder.layer.{l}."
            wb = rest.rsplit(".", 1)[1]
            if rest.startswith("norm1."):
                out[p + "layernorm_before." + wb] = v
            elif rest.startswith("norm2."):
                out[p + "layernorm_after." + wb] = v
            elif rest.startswith("attn.qkv."):
                q, kk, vv = v.chunk(3, dim=0)
                out[p + "attention.attention.query." + wb] = q
                out[p + "attention.attention.key." + wb] = kk
                out[p + "attention.attention.value." + wb] = vv
            elif rest.startswith("attn.proj."):
                out[p + "attention.output.dense." + wb] = v
            elif rest.startswith("mlp.fc1."):
                out[p + "intermediate.dense." + wb] = v
            elif rest.startswith("mlp.fc2."):
                out[p + "output.dense." + wb] = v
        elif k.startswith(vp + "norm."):
            out["dpt.layernorm." + k.rsplit(".", 1)[1]] = v
        elif k.startswith("pretrained.act_postprocess"):
            n = int(k[len("pretrained.act_postprocess")])
            rest = k.split(".", 2)[2]
            wb = rest.rsplit(".", 1)[1]
            if rest.startswith("0.project.0."):
                out[f"neck.reassemble_stage.readout_projects.{n - 1}.0.{wb}"] = v
            elif rest.startswith("3."):
                out[f"neck.reassemble_stage.layers.{n - 1}.projection.{wb}"] = v
            elif rest.startswith("4."):
                out[f"neck.reassemble_stage.layers.{n - 1}.resize.{wb}"] = v
        elif k.startswith("scratch.layer"):
            i = int(k[len("scratch.layer")])
            out[f"neck.convs.{i - 1}.weight"] = v
        elif k.startswith("scratch.refinenet"):
            i = int(k[len("scratch.refinenet")])
            rest = k.split(".", 2)[2]
            rest = (rest.replace("out_conv.", "projection.").replace("resConfUnit", "residual_layer")
                    .replace("conv1.", "convolution1.").replace("conv2.", "convolution2."))
            out[f"neck.fusion_stage.layers.{4 - i}.{rest}"] = v
        elif k.startswith("scratch.output_conv."):
            out["head.head." + k[len("scratch.output_conv."):]] = v
        else:
            raise KeyError(k)
    return out


def main_large():
    """DPT-Large (`backbone='vitl16_384'`): the restated timm `vit_large_patch16_384` + the reference-side
    reassemble stages (ConvTranspose2d k4s4 / k2s2, identity, conv3x3 s2) of oracle.dpt_forward_vitl16 against HuggingFace
    `DPTForDepthEstimation(is_hybrid=False)` on the same seeded weights."""
    from transformers import DPTConfig, DPTForDepthEstimation
    from oracle.dpt_oracle import dpt_forward_vitl16
    torch.set_num_threads(os.cpu_count())
    cfg = DPTConfig(is_hybrid=False, image_size=384, patch_size=16, hidden_size=1024, num_hidden_layers=24,
                    num_attention_heads=16, intermediate_size=4096, hidden_act="gelu", layer_norm_eps=1e-6, qkv_bias=True,
                    backbone_out_indices=[5, 11, 17, 23], neck_hidden_sizes=[256, 512, 1024, 1024],
                    reassemble_factors=[4, 2, 1, 0.5], readout_type="project", fusion_hidden_size=256, head_in_index=-1,
                    use_batch_norm_in_fusion_residual=False, add_projection=False, hidden_dropout_prob=0.0,
                    attention_probs_dropout_prob=0.0)
    hf = DPTForDepthEstimation(cfg).eval()
    worst = 0.0
    for seed in (5,):
        sd = random_state_dict(seed, 1, backbone="vitl16_384")
        missing, unexpected = hf.load_state_dict(to_hf(sd, hybrid=False), strict=False)
        unexpected = [u for u in unexpected if "fusion_stage.layers.0.residual_layer1" not in u]
        assert not missing and not unexpected, (missing, unexpected)
        x = synthetic_input(seed, 1, "depth")
        with torch.no_grad():
            y_hf = hf(pixel_values=x).predicted_depth
        y = dpt_forward_vitl16(sd, x)
        d = (y_hf - y).abs().max().item()
        worst = max(worst, d)
        print(f"DPT-Large seed={seed}: |HF(is_hybrid=False) - oracle.dpt_forward_vitl16|={d:.3e}  out std={y_hf.std():.3f}")
    assert worst < 5e-4, worst
    return worst


def main():
    from transformers import DPTConfig, DPTForDepthEstimation
    torch.set_num_threads(os.cpu_count())
    cfg = DPTConfig(is_hybrid=True, image_size=384, hidden_size=768, num_hidden_layers=12,
                    num_attention_heads=12, intermediate_size=3072, hidden_act="gelu",
                    layer_norm_eps=1e-6, qkv_bias=True, backbone_out_indices=[2, 5, 8, 11],
                    neck_hidden_sizes=[256, 512, 768, 768], backbone_featmap_shape=[1, 1024, 24, 24],
                    readout_type="project", fusion_hidden_size=256, head_in_index=-1,
                    use_batch_norm_in_fusion_residual=False, add_projection=False,
                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    hf = DPTForDepthEstimation(cfg).eval()
    worst = 0.0
    for seed in (0, 2):
        sd = random_state_dict(seed, 1)
        missing, unexpected = hf.load_state_dict(to_hf(sd), strict=False)
        # refinenet4.resConfUnit1 exists in the reference checkpoint but HF drops it
        unexpected = [u for u in unexpected if "fusion_stage.layers.0.residual_layer1" not in u]
        assert not missing and not unexpected, (missing, unexpected)
        x = synthetic_input(seed, 1, "depth")
        with torch.no_grad():
            y_hf = hf(pixel_values=x).predicted_depth
        y_timm = dpt_forward(sd, x, ws_form="timm04")
        y_hff = dpt_forward(sd, x, ws_form="hf")
        d1 = (y_hf - y_hff).abs().max().item()
        d2 = (y_hf - y_timm).abs().max().item()
        d3 = (y_timm - y_hff).abs().max().item()
        worst = max(worst, d1, d2)
        print(f"seed={seed}: |HF - oracle(ws=hf)|={d1:.3e}  |HF - oracle(ws=timm04)|={d2:.3e}  "
              f"|oracle timm04 - oracle hf|={d3:.3e}  out std={y_hf.std():.3f}")
    print("worst HF-vs-oracle difference:", worst)
    assert worst < 5e-4, worst


if __name__ == "__main__":
    main()
    main_large()
