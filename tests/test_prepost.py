"""Host logic of the GPU pre-processing (no GPU): Pillow's fixed-point bilinear coefficient tables, rebuilt in
libdptx.so, against (a) a numpy restatement and (b) Pillow itself through a full two-pass resample."""
import ctypes as C
import math

import numpy as np
import pytest
from PIL import Image

PREC = 22


def np_coeffs(in_size, out_size):
    scale = in_size / out_size
    fs = max(scale, 1.0)
    support = fs
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.array([max(0.0, 1.0 - abs((x + xmin - center + 0.5) / fs)) for x in range(xmax)])
        if w.sum() != 0:
            w = w / w.sum()
        kk[xx, :xmax] = [int(0.5 + v * (1 << PREC)) for v in w]
        bounds[xx] = (xmin, xmax)
    return bounds, kk, ksize


def lib_coeffs(lib, in_size, out_size):
    cap = out_size * (int(math.ceil(max(in_size / out_size, 1.0))) * 2 + 1)
    b = np.zeros((out_size, 2), np.int32)
    k = np.zeros(cap, np.int32)
    ks = C.c_int32()
    rc = lib.dptx_resample_coeffs(in_size, out_size, b.ctypes.data_as(C.POINTER(C.c_int32)),
                                  k.ctypes.data_as(C.POINTER(C.c_int32)), cap, C.byref(ks))
    assert rc == 0
    return b, k[: out_size * ks.value].reshape(out_size, ks.value), ks.value


def resample(a, bh, kh, bv, kv):
    H, W, Cc = a.shape
    tmp = np.zeros((H, len(bh), Cc), np.uint8)
    for x, (x0, n) in enumerate(bh):
        acc = (a[:, x0:x0 + n, :].astype(np.int64) * kh[x, :n][None, :, None]).sum(1) + (1 << (PREC - 1))
        tmp[:, x, :] = np.clip(acc >> PREC, 0, 255)
    out = np.zeros((len(bv), len(bh), Cc), np.uint8)
    for y, (y0, n) in enumerate(bv):
        acc = (tmp[y0:y0 + n].astype(np.int64) * kv[y, :n][:, None, None]).sum(0) + (1 << (PREC - 1))
        out[y] = np.clip(acc >> PREC, 0, 255)
    return out


@pytest.mark.parametrize("in_size,out_size", [(640, 480), (500, 640), (384, 384), (1000, 548), (250, 480), (777, 774), (3000, 384)])
def test_coefficient_tables_match_numpy_restatement(built_lib, in_size, out_size):
    from omnidata_amd.engine import load_library
    b, k, ks = lib_coeffs(load_library(), in_size, out_size)
    nb, nk, nks = np_coeffs(in_size, out_size)
    assert ks == nks and np.array_equal(b, nb) and np.array_equal(k, nk)


@pytest.mark.parametrize("h,w", [(512, 640), (300, 500), (1000, 700), (200, 250)])
def test_tables_reproduce_pillow_bit_exactly(built_lib, h, w):
    from omnidata_amd.engine import load_library
    lib = load_library()
    rng = np.random.default_rng(h * 1000 + w)
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ow, oh = (384, int(384 * h / w)) if w < h else (int(384 * w / h), 384)
    bh, kh, _ = lib_coeffs(lib, w, ow)
    bv, kv, _ = lib_coeffs(lib, h, oh)
    ref = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BILINEAR))
    assert np.array_equal(resample(a, bh, kh, bv, kv), ref)
