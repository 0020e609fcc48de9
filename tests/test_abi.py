"""C-ABI library: loads, exports every symbol include/dptx.h declares, and the host-side
weight folding / packing (no GPU needed: host-only handles, device_id = -1)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from omnidata_amd.weights import random_state_dict, state_dict_spec, is_unused

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dptx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dptx_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built_lib):
    lib = ctypes.CDLL(built_lib)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/dptx.h but not exported"


def test_python_binding_covers_header(built_lib):
    from omnidata_amd.engine import ABI, load_library
    load_library()
    assert sorted(n for n, _, _ in ABI) == declared_symbols()


def test_device_handle_fails_loudly_without_gpu(built_lib):
    from omnidata_amd.engine import Engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no device"):
        Engine(device_id=0)


def test_strict_loading_errors(built_lib):
    from omnidata_amd.engine import Engine
    e = Engine(num_channels=3, max_batch=1, device_id=None)
    sd = random_state_dict(0, 3)
    with pytest.raises(RuntimeError, match="unexpected key"):
        e.load_state_dict({"not.a.key": torch.zeros(1)})
    with pytest.raises(RuntimeError, match="shape mismatch"):
        e.load_state_dict({"scratch.output_conv.4.bias": torch.zeros(1)})  # C=3 expects [3]
    partial = {k: v for k, v in sd.items() if "blocks.7.attn.qkv" not in k}
    with pytest.raises(RuntimeError, match="missing tensors.*blocks.7.attn.qkv.weight"):
        e.load_state_dict(partial)
    # the tensors the forward never reads may be absent (and are ignored when present)
    e2 = Engine(num_channels=3, max_batch=1, device_id=None)
    e2.load_state_dict({k: v for k, v in sd.items() if not is_unused(k)})
    e3 = Engine(num_channels=3, max_batch=1, device_id=None)
    e3.load_state_dict(sd)
    assert np.array_equal(e2.export_packed_host(), e3.export_packed_host())
    with pytest.raises(RuntimeError, match="host-only"):
        e3._check(e3.lib.dptx_forward(e3.h, 1, 0, 1, 1, None), "forward")


def _blob_offsets(spec_items, dtype_bytes=2):
    """Re-derives the packed layout: a 256-byte layout header (engine.hip BlobHeader), then the entries in spec order,
    256-B aligned."""
    off, out = 256, {}
    for k, shape in spec_items:
        if is_unused(k):
            continue
        n = int(np.prod(shape))
        if k.endswith("stem.conv.weight"):
            b = 64 * 176 * 2
        elif k == "scratch.output_conv.4.weight":
            b = n * 4
        elif k.endswith(".weight") and len(shape) >= 2:
            b = n * 2
        else:
            b = n * 4
        out[k] = (off, b)
        off += (b + 255) // 256 * 256
        derived = []  # entries the engine computes at pack time (engine.hip R_DERIVED)
        if k.endswith("mlp.fc2.bias"):  # LayerNorm fold: column sums of the folded qkv / fc1 weights
            blk = k[:-len("mlp.fc2.bias")]
            derived = [(blk + "attn.qkv.lnsum", 3 * n), (blk + "mlp.fc1.lnsum", 4 * n),
                       # fp8 ViT (round 6): inverse weight scales of the e4m3 copies + column sums of the dequantised folded weights
                       (blk + "attn.qkv.f8scale", 3 * n), (blk + "mlp.fc1.f8scale", 4 * n), (blk + "mlp.fc2.f8scale", n),
                       (blk + "attn.qkv.lnsum8", 3 * n), (blk + "mlp.fc1.lnsum8", 4 * n)]
        elif k.endswith(".bias") and ("scratch.refinenet" in k or "scratch.output_conv.0." in k):
            derived = [(k[:-len("bias")] + "f8scale", n)]  # fp8 dtype: per-output-channel inverse weight scales
        for name, m in derived:
            out[name] = (off, m * 4)
            off += (m * 4 + 255) // 256 * 256
    return out, off


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_packed_weights_match_numpy_fold(built_lib, dtype):
    from omnidata_amd.engine import Engine
    from oracle.dpt_oracle import standardize_weight
    C = 1
    sd = random_state_dict(3, C)
    e = Engine(num_channels=C, max_batch=2, dtype=dtype, device_id=None, flags=1)  # DPTX_FLAG_NO_LN_FOLD: weights as loaded
    e.load_state_dict(sd)
    blob = e.export_packed_host()
    offs, total = _blob_offsets(state_dict_spec(C).items())
    assert total == blob.size == e.packed_bytes
    tdt = torch.bfloat16 if dtype == "bf16" else torch.float16
    # default engine: LayerNorm folded into qkv / fc1 -- W' = W diag(gamma) rounded, b' = b + W beta, lnsum = row sums of the
    # ROUNDED W' (what the MFMA multiplies the mean with)
    ef = Engine(num_channels=C, max_batch=2, dtype=dtype, device_id=None)
    ef.load_state_dict(sd)
    fblob = ef.export_packed_host()
    assert fblob.size == blob.size
    for wk, nk in (("attn.qkv", "norm1"), ("mlp.fc1", "norm2")):
        pre = "pretrained.model.blocks.7."
        W, g_, b_ = sd[pre + wk + ".weight"], sd[pre + nk + ".weight"], sd[pre + nk + ".bias"]
        o, nb = offs[pre + wk + ".weight"]
        got = torch.from_numpy(fblob[o:o + nb].copy()).view(tdt).float().reshape(W.shape)
        assert torch.equal(got, (W * g_[None, :]).to(tdt).float())
        o, nb = offs[pre + wk + ".lnsum"]
        cs = torch.from_numpy(fblob[o:o + nb].copy()).view(torch.float32)
        assert torch.allclose(cs.double(), got.double().sum(1), rtol=1e-6, atol=1e-6)
        o, nb = offs[pre + wk + ".bias"]
        bb = torch.from_numpy(fblob[o:o + nb].copy()).view(torch.float32)
        assert torch.allclose(bb.double(), sd[pre + wk + ".bias"].double() + W.double() @ b_.double(), rtol=1e-6, atol=1e-6)
    # everything the fold does not touch is identical in both blobs
    o, nb = offs["pretrained.model.blocks.7.attn.proj.weight"]
    assert np.array_equal(blob[o:o + nb], fblob[o:o + nb])

    def packed16(key, n):
        o, b = offs[key]
        return torch.from_numpy(blob[o:o + 2 * n].copy()).view(tdt).float()

    # linear: plain RNE conversion
    k = "pretrained.model.blocks.5.mlp.fc1.weight"
    assert torch.equal(packed16(k, sd[k].numel()), sd[k].to(tdt).float().flatten())
    # plain conv: OIHW -> O,kh,kw,I
    k = "scratch.refinenet2.resConfUnit1.conv2.weight"
    assert torch.equal(packed16(k, sd[k].numel()), sd[k].permute(0, 2, 3, 1).to(tdt).float().flatten())
    # StdConv (3x3 and 1x1): standardised (timm 0.4.x form, eps 1e-8), then re-laid out
    for k in ("pretrained.model.patch_embed.backbone.stages.1.blocks.0.conv2.weight",
              "pretrained.model.patch_embed.backbone.stages.2.blocks.3.conv3.weight"):
        ref = standardize_weight(sd[k].double(), 1e-8, "timm04").permute(0, 2, 3, 1).flatten()
        got = packed16(k, sd[k].numel()).double()
        tol = 2.0 ** (-8 if dtype == "bf16" else -11) * ref.abs().clamp_min(1e-3)  # <= 1 ulp: fold done in double
        assert bool(((got - ref).abs() <= tol).all())
        assert (got - ref.float().to(tdt).double()).abs().max() <= 2.0 ** (-7 if dtype == "bf16" else -10) * ref.abs().max()
    # stem: [64][176], k = (c*7 + ky)*8 + kx with kx padded 7 -> 8 and K padded 168 -> 176 (zeros)
    k = "pretrained.model.patch_embed.backbone.stem.conv.weight"
    got = packed16(k, 64 * 176).reshape(64, 176)
    ref = standardize_weight(sd[k].double(), 1e-8, "timm04")                     # [64,3,7,7] (o,c,ky,kx)
    g4 = got[:, :168].reshape(64, 3, 7, 8)
    assert torch.all(got[:, 168:] == 0) and torch.all(g4[..., 7] == 0)
    assert (g4[..., :7].double() - ref).abs().max() <= 2.0 ** (-7 if dtype == "bf16" else -10) * ref.abs().max()
    # fp32 vectors verbatim
    for k in ("pretrained.model.pos_embed", "scratch.output_conv.4.weight", "pretrained.model.blocks.0.norm1.bias"):
        o, b = offs[k]
        assert torch.equal(torch.from_numpy(blob[o:o + b].copy()).view(torch.float32), sd[k].flatten())


def test_fp16_conversion_edge_cases(built_lib):
    """Host fp32->fp16 RNE (engine.hip f32_to_fp16) against torch, incl. subnormals/overflow."""
    from omnidata_amd.engine import Engine
    vals = torch.tensor([0.0, -0.0, 1.0, -1.0, 65504.0, 65519.9, 65520.0, 1e6, 6.1035e-5, 6.0e-5, 5.96e-8, 2.98e-8,
                         2.9e-8, 1e-9, 0.333333, 1.0009765625, 1.00048828125, 1.00146484375, -3.14159, 12345.678])
    sd = random_state_dict(0, 3)
    k = "pretrained.model.blocks.0.attn.proj.weight"
    w = sd[k].clone()
    w.view(-1)[: vals.numel()] = vals
    rnd = torch.randn(4096) * torch.logspace(-9, 5, 4096)
    w.view(-1)[100:100 + 4096] = rnd
    sd[k] = w
    e = Engine(num_channels=3, max_batch=1, dtype="fp16", device_id=None)
    e.load_state_dict(sd)
    blob = e.export_packed_host()
    offs, _ = _blob_offsets(state_dict_spec(3).items())
    o, b = offs[k]
    got = torch.from_numpy(blob[o:o + b].copy()).view(torch.float16)
    ref = w.flatten().to(torch.float16)
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16))


def test_bf16x3_packing_has_hi_and_lo_planes(built_lib):
    """bf16x3 blob = [hi blob | lo blob]: hi is exactly the bf16 blob, lo = bf16(w - hi)."""
    from omnidata_amd.engine import Engine
    sd = random_state_dict(4, 3)
    e1 = Engine(num_channels=3, max_batch=1, dtype="bf16", device_id=None, flags=1)  # no LayerNorm fold: bf16x3 never folds
    e1.load_state_dict(sd)
    e3 = Engine(num_channels=3, max_batch=1, dtype="bf16x3", device_id=None)
    e3.load_state_dict(sd)
    b1, b3 = e1.export_packed_host(), e3.export_packed_host()
    assert b3.size == 2 * b1.size == e3.packed_bytes
    assert np.array_equal(b3[256: b1.size], b1[256:])   # behind the layout header (dtype, sizes)
    offs, _ = _blob_offsets(state_dict_spec(3).items())
    k = "pretrained.model.blocks.2.mlp.fc2.weight"
    o, b = offs[k]
    lo = torch.from_numpy(b3[b1.size + o: b1.size + o + b].copy()).view(torch.bfloat16).float()
    w = sd[k].flatten()
    hi = w.to(torch.bfloat16).float()
    assert torch.equal(lo, (w - hi).to(torch.bfloat16).float())
    assert (w - hi - lo).abs().max() <= 2.0 ** -16 * w.abs().max()   # 16 significand bits
    # fp32 vectors live only in the hi half
    o, b = offs["pretrained.model.pos_embed"]
    assert not b3[b1.size + o: b1.size + o + b].any()


def test_vitl16_host_packing(built_lib):
    """DPT-Large behind the same ABI (dptx_config.backbone): strict loading of the vitl16 key set, and the ConvTranspose2d
    weights of reassemble stages 1 / 2 packed as the GEMM operand [(dy*k + dx)*Cout + co][ci] (engine.hip R_DECONV)."""
    from omnidata_amd.engine import Engine
    sd = random_state_dict(3, 1, backbone="vitl16_384")
    e = Engine(num_channels=1, max_batch=1, dtype="bf16", device_id=None, backbone="vitl16_384")
    with pytest.raises(RuntimeError, match="unexpected key"):
        e.load_state_dict({"pretrained.model.patch_embed.backbone.stem.conv.weight": torch.zeros(64, 3, 7, 7)})  # a hybrid key
    e.load_state_dict(sd)
    blob = e.export_packed_host().tobytes()
    for n, k in ((1, 4), (2, 2)):
        w = sd[f"pretrained.act_postprocess{n}.4.weight"]            # [Cin, Cout, k, k]
        want = w.permute(2, 3, 1, 0).reshape(k * k * w.shape[1], w.shape[0]).contiguous()   # [(dy, dx, co)][ci]
        pos = blob.find(want.to(torch.bfloat16).view(torch.int16).numpy().tobytes())
        assert pos >= 0 and pos % 256 == 0, n
        b = sd[f"pretrained.act_postprocess{n}.4.bias"]
        assert blob.find(b.repeat(k * k).numpy().tobytes()) >= 0     # the bias, once per (dy, dx)
    e.close()
    with pytest.raises(RuntimeError):
        Engine(num_channels=3, max_batch=1, device_id=None, backbone="vitl16_384", dual=True)   # the dual-task model is the hybrid


def test_fp8_vit_host_packing(built_lib):
    """Round 6, DPTX_FLAG_FP8_VIT: the fp8 blob carries e4m3 copies of qkv / fc1 / fc2 (second plane, byte offset / 2) of the
    weights the bf16 GEMM multiplies -- qkv / fc1 folded with the LayerNorm's gamma --, quantised per output channel after a
    power-of-two scale into (224, 448], the inverse scales ('f8scale') and the column sums of the DEQUANTISED copy ('lnsum8':
    the fold's mean term has to cancel against what the e4m3 MFMA multiplies).  Checked against numpy on the host-only handle."""
    from omnidata_amd.engine import Engine
    C = 3
    sd = random_state_dict(5, C)
    e = Engine(num_channels=C, max_batch=1, dtype="fp8", device_id=None, flags=32)
    e.load_state_dict(sd)
    blob = e.export_packed_host()
    offs, single = _blob_offsets(state_dict_spec(C).items())
    assert blob.size == 2 * single == e.packed_bytes
    f8 = torch.float8_e4m3fn
    for blk in (0, 7):
        p = f"pretrained.model.blocks.{blk}."
        for name, gkey in (("attn.qkv", "norm1.weight"), ("mlp.fc1", "norm2.weight"), ("mlp.fc2", None)):
            W = sd[p + name + ".weight"].double()
            Wf = W * sd[p + gkey].double()[None, :] if gkey else W
            N, K = Wf.shape
            mx = Wf.abs().amax(1)
            sc = torch.pow(2.0, torch.floor(torch.log2(448.0 / mx)))
            want8 = (Wf * sc[:, None]).float().to(f8)
            off, nbytes = offs[p + name + ".weight"]
            got8 = torch.from_numpy(blob[single + off // 2: single + off // 2 + N * K].copy()).view(f8).reshape(N, K)
            # the engine folds in fp32 (W * gamma) before scaling: a product that lands on a rounding boundary may go either way
            same = (got8.view(torch.uint8) == want8.view(torch.uint8)).float().mean().item()
            assert same > 0.995, (name, same)
            assert ((got8.float() - want8.float()).abs() <= torch.exp2(torch.floor(torch.log2(want8.float().abs().clamp_min(2.0 ** -6))) - 3) * 1.001 + 1e-9).all()
            so, _ = offs[p + name + ".f8scale"]
            inv = torch.from_numpy(blob[so: so + 4 * N].copy()).view(torch.float32)
            assert torch.equal(inv, (1.0 / sc).float())
            assert ((got8.float().abs().amax(1) > 224 * 0.93) & (got8.float().abs().amax(1) <= 448)).all()   # the channel fills the e4m3 range
            if gkey:
                lo, _ = offs[p + name + ".lnsum8"]
                cs = torch.from_numpy(blob[lo: lo + 4 * N].copy()).view(torch.float32)
                want_cs = (got8.double().sum(1) / sc).float()
                assert torch.allclose(cs, want_cs, rtol=1e-6, atol=1e-7)
