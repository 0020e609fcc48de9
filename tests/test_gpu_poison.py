"""Arena-poison invariance: a forward must not read an arena byte it has not written itself.

For every dtype, single- and dual-task, one stream and two: the reference result, then the whole arena (both planes) is set
to 0xFF (NaN in fp32 / bf16 / fp16 / e4m3), to 0x00 and to the leftovers of a forward at another batch size -- the output
bits must not change.  A kernel whose result depends on prior memory contents shows up here deterministically (NaN
propagates), whichever box runs it.  pytest -m gpu."""
import pytest
import torch

from omnidata_amd.engine import Engine
from omnidata_amd.weights import random_dual_state_dict, random_state_dict, synthetic_input

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [  # dtype, dual, B, streams
    ("bf16", False, 3, 2), ("bf16", True, 3, 2), ("bf16", False, 1, 1),
    ("fp16", False, 3, 2),
    ("mixed", False, 3, 2), ("mixed", True, 3, 2), ("mixed", False, 1, 1),
    ("fp16x3", False, 2, 2), ("bf16x3", False, 2, 1),
    ("fp8", False, 3, 2), ("fp8", True, 3, 2), ("fp8", True, 3, 1), ("fp8", False, 1, 1),
    ("fp8all", True, 3, 2), ("fp8all", False, 3, 2),   # DPTX_FLAG_FP8_ALL: all 19 eligible decoder convolutions on e4m3
    ("fp8vit", False, 3, 2), ("fp8vit", True, 3, 1),   # DPTX_FLAG_FP8_VIT (round 6): qkv / fc1 / fc2 on e4m3, e4m3 copies of the token stream
]


def _fwd(eng, x, dual):
    if dual:
        yn, yd = eng.forward_dual(x)
        torch.cuda.synchronize()
        return torch.cat([yn.flatten(1), yd.flatten(1)], dim=1).clone()
    y = eng.forward(x)
    torch.cuda.synchronize()
    return y.flatten(1).clone()


@pytest.mark.parametrize("dtype,dual,B,streams", CASES)
def test_forward_does_not_depend_on_prior_arena_contents(dtype, dual, B, streams):
    flags = {"fp8all": 16, "fp8vit": 32}.get(dtype, 0)
    dtype = "fp8" if dtype in ("fp8all", "fp8vit") else dtype
    eng = Engine(num_channels=3, max_batch=B, dtype=dtype, device_id=0, dual=dual, streams=streams, flags=flags)
    eng.load_state_dict(random_dual_state_dict(3) if dual else random_state_dict(3, 3))
    x = synthetic_input(11, B, "normal").to(DEV)
    if dtype == "fp8":
        eng.calibrate_fp8(x)   # (its own forward uses the whole-batch layout and a bf16 decoder: more leftovers)
    ref = _fwd(eng, x, dual)
    assert torch.isfinite(ref).all()
    for pattern in (0xFF, 0x00, 0x7F):
        eng.arena_fill(pattern)
        y = _fwd(eng, x, dual)
        n = int(((y != ref) | (torch.isnan(y) != torch.isnan(ref))).sum())
        assert n == 0, f"{n} output elements changed after the arena was filled with 0x{pattern:02X} ({int(torch.isnan(y).sum())} NaN)"
    if B > 1:  # leftovers of a forward at another batch size (other sub-batch split, other tile shapes)
        _fwd(eng, x[:1], dual)
        y = _fwd(eng, x, dual)
        assert torch.equal(y, ref)
        y1 = _fwd(eng, x[:1], dual)
        assert torch.equal(y1[0], ref[0]), "batch invariance"
    eng.close()


def test_packed_blob_of_another_layout_is_refused():
    """the packed layout depends on more than the size (LayerNorm fold, dtype ...); the blob carries a layout
    header and dptx_import_packed_device refuses one that this handle would not have produced itself."""
    sd = random_state_dict(3, 3)
    src = Engine(num_channels=3, max_batch=1, dtype="bf16", device_id=0, flags=1)  # DPTX_FLAG_NO_LN_FOLD: unfolded qkv / fc1
    src.load_state_dict(sd)
    blob = src.export_packed()
    dst = Engine(num_channels=3, max_batch=1, dtype="bf16", device_id=0)
    assert dst.packed_bytes == src.packed_bytes
    with pytest.raises(RuntimeError, match="another layout"):
        dst.import_packed(blob)
    same = Engine(num_channels=3, max_batch=1, dtype="bf16", device_id=0, flags=1)
    same.import_packed(blob)
    x = synthetic_input(1, 1, "normal").to(DEV)
    assert torch.equal(same.forward(x), src.forward(x))
    for e in (src, dst, same):
        e.close()


def test_range_flag_is_sticky_and_resets():
    """include/dptx.h dptx_range_status: clean weights leave the flag clear; weights whose decoder activations exceed the fp16
    range set it in the fp16-plane dtypes (not in bf16 planes), it stays set until it is read with reset."""
    sd = random_state_dict(0, 3)
    big = {k: v.clone() for k, v in sd.items()}
    for k in big:
        if k.startswith("scratch.layer") and k.endswith("_rn.weight"):
            big[k] *= 1.0e8
        if k.startswith("scratch.refinenet") and k.endswith(".bias"):
            big[k] *= 1.0e8
    big["scratch.output_conv.0.weight"] /= 1.0e8
    x = synthetic_input(0, 2, "normal").to(DEV)
    for dtype, weights, want in (("mixed", sd, False), ("mixed", big, True), ("fp16", big, True), ("bf16x3", big, False), ("bf16", big, False)):
        eng = Engine(num_channels=3, max_batch=2, dtype=dtype, device_id=0)
        eng.load_state_dict(weights)
        eng.forward(x)
        assert eng.range_overflowed(reset=False) == want, (dtype, want)
        assert eng.range_overflowed(reset=True) == want      # sticky until reset
        assert eng.range_overflowed(reset=True) is False
        eng.close()
