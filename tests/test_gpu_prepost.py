"""GPU pre/post-processing against the reference's own host path (PIL / torch): the uint8 resize is
integer work and must be bit-exact.  pytest -m gpu."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from PIL import Image

from omnidata_amd import preprocess as pp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,w,mode", [(512, 640, "RGB"), (300, 500, "RGB"), (1000, 701, "RGB"), (384, 384, "RGB"), (400, 400, "L"),
                                      (385, 777, "RGB"), (2160, 3840, "RGB"), (384, 1000, "L")])
@pytest.mark.parametrize("task", ["normal", "depth"])
def test_preprocess_bit_exact_vs_pil_path(h, w, mode, task):
    rng = np.random.default_rng(h + w)
    shape = (h, w, 3) if mode == "RGB" else (h, w)
    img = Image.fromarray(rng.integers(0, 256, shape, dtype=np.uint8))
    ref = pp.image_to_input(img, task)                 # PIL Resize/CenterCrop/ToTensor(/Normalize): the reference path
    got = pp.image_to_input_gpu(img, task).cpu()
    assert got.shape == ref.shape == (1, 3, 384, 384)
    assert torch.equal(got, ref)


def test_postprocess_matches_reference_ops():
    y = (torch.rand(3, 384, 384) * 1.4 - 0.2).cuda()
    got = pp.normal_to_u8_gpu(y).cpu().numpy()
    assert np.array_equal(got, np.asarray(pp.normal_to_pil(y.cpu())))
    d = (torch.rand(384, 384) * 1.2 - 0.1).cuda()
    ref = 1 - F.interpolate(d.cpu().clamp(0, 1)[None, None], (512, 512), mode="bicubic").clamp(0, 1)[0, 0]
    # demo.py clamps the model output first (line 140), then interpolates, clamps again and flips
    got = pp.depth_to_512_gpu(d.clamp(0, 1)).cpu()
    assert (got - ref).abs().max() < 2e-6
