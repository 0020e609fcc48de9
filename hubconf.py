"""torch.hub entry points mirroring the reference's hub surface (README.md:18-32,
omnidata_tools/torch/README.md:41-57; the reference's own hubconf lives in the external repo
alexsax/omnidata_models):

    model = torch.hub.load('/root/repo', 'surface_normal_dpt_hybrid_384', source='local')
    model = torch.hub.load('/root/repo', 'depth_dpt_hybrid_384', source='local')
    model = torch.hub.load('/root/repo', 'dpt_hybrid_384', source='local', pretrained=False, task='normal')

Normal models take 384x384 RGB in [0,1], depth models in [-1,1].  The returned nn.Module runs
its forward in libdptx.so on an MI355X.  Engine kwargs: max_batch, dtype -- default 'mixed', the mode that
matches the reference's fp32 forward within 1e-3; 'fp16x3' / 'bf16x3' are reference-grade, 'bf16' / 'fp16' are
~1.6x faster single-pass throughput modes that do NOT meet 1e-3 ('fp8': a lossy decoder variant of 'bf16' that is not faster
than it on the boxes measured -- omnidata_amd/model.py DPTDepthModel).
Inputs of other sizes (H, W multiples of 32) are accepted as the reference accepts them (resized pos_embed).
`dual_dpt_hybrid_384` (not in the reference) returns normals and depth from one encoder pass.
"""
import os

dependencies = ["torch"]

_CKPT = {"normal": "omnidata_dpt_normal_v2.ckpt", "depth": "omnidata_dpt_depth_v2.ckpt"}


def _find_ckpt(task, weights=None):
    if weights:
        return weights
    for d in (os.environ.get("OMNIDATA_PRETRAINED_DIR"), "./pretrained_models", os.path.join(os.path.dirname(__file__), "pretrained_models")):
        if d and os.path.isfile(os.path.join(d, _CKPT[task])):
            return os.path.join(d, _CKPT[task])
    raise FileNotFoundError(
        f"pretrained=True needs {_CKPT[task]} (tools/download_*_models.sh in the reference) in ./pretrained_models or "
        "$OMNIDATA_PRETRAINED_DIR; this build environment has no network. Use pretrained=False for seeded synthetic weights.")


def dpt_hybrid_384(pretrained=False, task="normal", weights=None, **kwargs):
    from omnidata_amd.model import build_model
    return build_model(task, weights=_find_ckpt(task, weights) if pretrained else weights, **kwargs)


def surface_normal_dpt_hybrid_384(pretrained=True, **kwargs):
    return dpt_hybrid_384(pretrained=pretrained, task="normal", **kwargs)


def depth_dpt_hybrid_384(pretrained=True, **kwargs):
    return dpt_hybrid_384(pretrained=pretrained, task="depth", **kwargs)


def dual_dpt_hybrid_384(pretrained=False, backbone="normal", normal_weights=None, depth_weights=None, **kwargs):
    """Not in the reference's hub list: normals + depth from ONE encoder pass (omnidata_amd.model.DPTDualTaskModel).
    With pretrained=True the two v2 checkpoints are read and `backbone` ('normal'|'depth') donates `pretrained.*`."""
    from omnidata_amd.model import DPTDualTaskModel
    from omnidata_amd.weights import read_checkpoint
    if not pretrained and not (normal_weights and depth_weights):
        return DPTDualTaskModel(**kwargs).eval()
    n = read_checkpoint(_find_ckpt("normal", normal_weights))
    d = read_checkpoint(_find_ckpt("depth", depth_weights))
    return DPTDualTaskModel.from_single_task(n, d, backbone=backbone, **kwargs).eval()
