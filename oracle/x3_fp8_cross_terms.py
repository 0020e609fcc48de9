"""Test infrastructure (oracle/): CPU emulation of a cheaper extended-precision product for the parity mode (VERDICT r4 item 6,
DESIGN.md section 9 item 2) -- the op-level probe against fp64 that has to hold BEFORE any kernel is written.

The parity mode's 3-MFMA product of fp16 hi/lo planes is   a w  ~  a_hi w_hi + a_hi w_lo + a_lo w_hi   (three bf16-rate MFMAs).
The two cross terms are 2^-11 of the leading term, so they need only a few significant bits themselves.  Candidate: keep
a_hi w_hi on the fp16 MFMA and run BOTH cross terms as ONE block-scaled fp8 MFMA over a K-concatenated operand
    [a_hi | a_lo] . [w_lo | w_hi]      (v_mfma_scale_f32_32x32x64_f8f6f4: e4m3 elements, one E8M0 scale per 32 consecutive k)
which costs one bf16-MFMA equivalent (twice the K at twice the rate): 2 instead of 3 per product; with fp6 (e2m3, the fp4 rate)
1.5.  The question this script answers: how much of the lo planes' benefit survives quantising the cross-term operands to
MX-fp8 / MX-fp6?

    python -m oracle.x3_fp8_cross_terms        (numpy only; prints a table, writes nothing)

Shapes and value distributions follow the layers the parity mode spends three MFMAs on (decoder convolutions: K = 2304,
activations ~ post-ReLU half-normal with a heavy tail, weights ~ N(0, 1/K))."""
import numpy as np


def to_fp16(x):
    return x.astype(np.float16).astype(np.float64)


def quant_mx(x, mant_bits, emax, block=32):
    """MX block quantisation along the last axis: one power-of-two scale per `block` elements (E8M0) putting the block's
    max |x| just inside the element format's range, elements rounded to a float with `mant_bits` explicit mantissa bits and
    normal exponents down to 2^(emax - span): e4m3: mant 3, emax 8 (448 = 1.75 * 2^8); e2m3 (fp6): mant 3, emax 2 (7.5)."""
    shp = x.shape
    xb = x.reshape(-1, block)
    amax = np.abs(xb).max(axis=1, keepdims=True)
    amax = np.where(amax == 0, 1.0, amax)
    # scale so that amax lands in [2^emax, 2^(emax+1)) (top binade of the element format)
    s = 2.0 ** (emax - np.floor(np.log2(amax)))
    y = xb * s
    span = {3: {8: 15, 2: 3}}[mant_bits][emax]      # number of normal binades below the top one
    e = np.floor(np.log2(np.maximum(np.abs(y), 1e-300)))
    e = np.clip(e, emax - span, emax)               # below the smallest normal binade: subnormal spacing of that binade
    q = 2.0 ** (e - mant_bits)
    yq = np.round(y / q) * q
    top = (2.0 - 2.0 ** (-mant_bits)) * 2.0 ** emax
    yq = np.clip(yq, -top, top)
    return (yq / s).reshape(shp)


def quant_e5m2(x16, div):
    """What v_cvt_scalef32_pk_bf8_f16 does to an fp16 plane (semantics pinned on the hardware: profiles/r06_bf8_probe.txt): e5m2 = the upper byte of an fp16, so
    the conversion is fp16(x / div) rounded to nearest even on two mantissa bits, subnormals kept (spacing 2^-16).  No block
    scale: hi planes are divided by 2 (65504 / 2 stays below e5m2's largest finite value 57344), lo planes -- at most 2^-11 of
    their hi plane -- are multiplied by 2^10, which also lifts them out of the format's subnormal range; the MFMA's E8M0 scale
    operand takes the 2^-9 back out.  Returns the de-scaled values as float64."""
    y = (x16.astype(np.float64) / div).astype(np.float16)
    u = y.view(np.uint16).astype(np.uint32)
    r = ((u + 0x7F + ((u >> 8) & 1)) & 0xFF00).astype(np.uint16)
    return r.view(np.float16).astype(np.float64) * div


def study(M=512, N=256, K=2304, seed=0):
    rng = np.random.default_rng(seed)
    a = np.maximum(rng.standard_normal((M, K)), 0.0) * (1.0 + 3.0 * (rng.random((M, K)) < 0.01))   # post-ReLU, 1 % outliers
    w = rng.standard_normal((N, K)) / np.sqrt(K)
    ref = a @ w.T                                        # fp64
    a_hi, w_hi = to_fp16(a), to_fp16(w)
    a_lo, w_lo = to_fp16(a - a_hi), to_fp16(w - w_hi)
    scale = np.sqrt((ref ** 2).mean())

    def err(y):
        d = y - ref
        return np.sqrt((d ** 2).mean()) / scale, np.abs(d).max() / scale

    rows = []
    rows.append(("1 MFMA  : a_hi w_hi (fp16 single pass)", 1.0, a_hi @ w_hi.T))
    rows.append(("2 MFMAs : a_hi (w_hi + w_lo) (weights exact, engine's XT = 2)", 2.0, a_hi @ (w_hi + w_lo).T))
    rows.append(("3 MFMAs : + a_lo w_hi (the parity mode's product)", 3.0, a_hi @ (w_hi + w_lo).T + a_lo @ w_hi.T))
    for name, mant, emax, cost in (("MX-fp8 e4m3", 3, 8, 2.0), ("MX-fp6 e2m3", 3, 2, 1.5)):
        q = lambda t: quant_mx(t, mant, emax)   # noqa: E731
        cross = q(a_hi) @ q(w_lo).T + q(a_lo) @ q(w_hi).T
        rows.append((f"1 MFMA + cross terms on {name} (block 32)", cost, a_hi @ w_hi.T + cross))
        cross_w = q(a_hi) @ q(w_lo).T
        rows.append((f"1 MFMA + ONLY a_hi w_lo on {name} (the XT = 2 analogue)", 1.0 + (cost - 1.0) / 2, a_hi @ w_hi.T + cross_w))
    # round 6, the form that was BUILT AND MEASURED (profiles/r06_x3_bf8_cross_attempt.patch; not shipped: the in-register
    # conversions cost more than the MFMA passes they save, profiles/r06_experiments.md section 7): e5m2 operands straight from
    # the fp16 planes, fixed scales, no block maxima
    h16 = lambda t: t.astype(np.float16)   # noqa: E731
    cross = quant_e5m2(h16(a_hi), 2.0) @ quant_e5m2(h16(w_lo), 2.0 ** -10).T + quant_e5m2(h16(a_lo), 2.0 ** -10) @ quant_e5m2(h16(w_hi), 2.0).T
    rows.append(("1 MFMA + cross terms on e5m2, fixed scales (round 6's kernel attempt)", 2.0, a_hi @ w_hi.T + cross))
    print(f"product [{M} x {K}] . [{N} x {K}]^T, post-ReLU activations with 1 % outliers, N(0, 1/K) weights; errors relative to rms(output)")
    print(f"{'form':72s} {'MFMA-equivalents':>16s} {'rms error':>12s} {'max error':>12s}")
    for name, cost, y in rows:
        r, m = err(y)
        print(f"{name:72s} {cost:16.2f} {r:12.3e} {m:12.3e}")


if __name__ == "__main__":
    study()
    study(K=256, seed=1)
