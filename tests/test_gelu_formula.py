"""The GEMM epilogues evaluate erf-GELU (timm Mlp / ProjectReadout: nn.GELU(), the exact erf form) with erfc by Abramowitz &
Stegun 7.1.28 in fp32 (omnidata_amd/csrc/common.h gelu_erf2).  This restates the device function step by step in numpy
fp32 (same operations, same order; fused multiply-adds emulated in float64 and rounded once) and pins its error against the
exact function: the kernel's own GPU test compares a whole GEMM + GELU with torch, this one isolates the formula."""
import math

import numpy as np
from scipy.special import erf

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * np.asarray(b, dtype=np.float64) + np.asarray(c, dtype=np.float64)).astype(f32)


def gelu_device(x):
    x = x.astype(f32)
    h = (x * f32(0.5)).astype(f32)
    ha = np.abs(h)
    z = (ha * f32(1.41421356237309504880)).astype(f32)
    q = fma(z, f32(0.0000430638), f32(0.0002765672))
    for c in (0.0001520143, 0.0092705272, 0.0422820123, 0.0705230784, 1.0):
        q = fma(q, z, f32(c))
    with np.errstate(over="ignore"):
        for _ in range(4):
            q = (q * q).astype(f32)
        r = (f32(1.0) / q).astype(f32)  # v_rcp_f32 is within 1 ulp of this
    return fma(-ha, r, (h + ha).astype(f32))


def test_gelu_formula_error():
    x = np.linspace(-12.0, 12.0, 2_000_001).astype(f32)
    exact = 0.5 * x.astype(np.float64) * (1.0 + erf(x.astype(np.float64) / math.sqrt(2.0)))
    got = gelu_device(x).astype(np.float64)
    err = np.abs(got - exact)
    assert err.max() < 1.0e-6, err.max()          # measured 7.1e-7 (7.1.26, used in rounds 1-2: 4.7e-7)
    big = np.abs(exact) > 1e-3
    assert (err[big] / np.abs(exact[big])).max() < 4e-4   # relative: the negative tail, where gelu itself is ~1e-3


def test_gelu_formula_limits():
    x = np.array([0.0, -0.0, 30.0, -30.0, 1e4, -1e4, 3e38, -3e38], dtype=f32)
    y = gelu_device(x)
    assert y[0] == 0.0 and y[1] == 0.0
    assert y[2] == 30.0 and y[4] == 1e4 and y[6] == f32(3e38)   # p^16 overflows to inf, its reciprocal is erfc's limit 0
    assert y[3] == 0.0 and y[5] == 0.0 and y[7] == 0.0 and not np.isnan(y).any()
