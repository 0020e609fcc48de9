"""Co-residency regression.

Round 3's multi-stream "nondeterminism" was one instruction form: hipcc's packed LayerNorm-fold arithmetic
(`v_pk_fma_f32 ... op_sel:[0,1,0]`) returned a zero product in its low lane, a few times per thousand launches, whenever the
STEM CONVOLUTION of another stream shared the CU.  The fix (scalar fmas, gemm_impl.h ln_fold_fma) is pinned statically by
tests/test_build_quality.py (ISA scan for that form); THIS file is the dynamic half -- the op-level victim x aggressor loop
of tools/gpu/r4_micro.py as a test, against every kernel form that can serve a LayerNorm-fold consumer (qkv / fc1):

    staged        gemm_glds_kernel, 128x128 tile, block-wide LDS epilogue            (small M)
    wave-private  gemm_pp_kernel, 256x256 tile, per-wave LDS epilogue                (large M, debug flag 1: no direct form)
    direct        gemm_pp_kernel<DIRECT>, epilogue_direct: stores from the registers (large M, the forward's form)

in bf16 and fp16 (the `mixed` dtype's ViT blocks are the fp16 ones; fp8's are the bf16 ones; the 3-MFMA dtypes have no fold),
plus a GELU-epilogue GEMM WITHOUT the fold -- the remaining packed-fp32 user (GELU polynomial, `op_sel_hi` splats).  Each
victim runs back to back into distinct outputs while the stem convolution loops on a second stream; every output must equal
the one computed alone, bit for bit.

Positive control: a library built with -DDPTX_LN_PACKED_FMA (round 3's arithmetic; `python tests/test_gpu_coresidency.py
--build-control` builds omnidata_amd/libdptx_lnpk.so) must SHOW the effect in the same loop -- otherwise a green run proves
nothing.  It runs in a child process (DPTX_LIB is read at load time); without that library the control is skipped.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CONTROL_LIB = os.path.join(ROOT, "omnidata_amd", "libdptx_lnpk.so")
CONTROL_FLAGS = "-DDPTX_LN_PACKED_FMA -Xclang -target-feature -Xclang +packed-fp32-ops"


def _lib():
    from omnidata_amd.engine import load_library
    return load_library()


def _ptr(t):
    return None if t is None else t.data_ptr()


def _st():
    return torch.cuda.current_stream().cuda_stream


class Victim:
    """One GEMM of a ViT block's shape: LayerNorm-fold consumer (ln=True: qkv / fc1) or plain bias (+ GELU) epilogue."""

    def __init__(self, dtype, M, N, K=768, act=0, ln=True, flags=0):
        from omnidata_amd.engine import DTYPES
        g = torch.Generator().manual_seed(1)
        self.tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
        self.mode, self.M, self.N, self.K, self.act, self.ln, self.flags = DTYPES[dtype], M, N, K, act, ln, flags
        self.A = (torch.randn(M, K, generator=g) * 2.0 + 0.3).to(self.tdt).to(DEV)
        xa = self.A.float()
        self.W = (torch.randn(N, K, generator=g) / K ** 0.5).to(self.tdt).to(DEV)
        self.bias = torch.randn(N, generator=g).to(DEV)
        stats = torch.zeros(M, 8, 2, device=DEV)
        for b in range(K // 128):
            blk = xa[:, b * 128:(b + 1) * 128]
            stats[:, b, 0] = blk.sum(1)
            stats[:, b, 1] = (blk * blk).sum(1)
        self.stats = stats.contiguous()
        self.colsum = self.W.float().sum(1).contiguous()

    def launch(self, out):
        lib = _lib()
        lib.dptx_debug_set_gemm_flags(self.flags)
        try:
            if self.ln:
                rc = lib.dptx_op_gemm_ln(self.mode, _ptr(self.A), _ptr(self.W), _ptr(self.bias), _ptr(out), self.M, self.N, self.K,
                                         self.act, _ptr(self.stats), _ptr(self.colsum), self.K // 128, 1e-6, _st())
            else:
                rc = lib.dptx_op_gemm(self.mode, _ptr(self.A), _ptr(self.W), _ptr(self.bias), None, _ptr(out), self.M, self.N, self.K,
                                      self.act, 0, 0, 0, _st())
        finally:
            lib.dptx_debug_set_gemm_flags(0)
        assert rc == 0, rc

    def empty(self):
        return torch.empty(self.M, self.N, dtype=self.tdt, device=DEV)


def stem_aggressor(dtype, images=4):
    from omnidata_amd.engine import DTYPES
    tdt = torch.float16 if dtype == "fp16" else torch.bfloat16
    x = torch.rand(images, 3, 384, 384, device=DEV)
    Wt = (torch.randn(64, 176) * 0.1).to(tdt).to(DEV)
    y = torch.empty(images, 192, 192, 64, dtype=tdt, device=DEV)

    def run():
        assert _lib().dptx_op_stem_conv(DTYPES[dtype], _ptr(x), _ptr(Wt), _ptr(y), images, 384, 384, _st()) == 0
    return run


def victim_next_to_aggressor(victim, aggr, launches, per_round, aggr_per_round):
    """Number of victim launches (of `launches`) whose output differs from the one computed alone."""
    ref = victim.empty()
    victim.launch(ref)
    torch.cuda.synchronize()
    outs = [victim.empty() for _ in range(per_round)]
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    bad = 0
    for _ in range((launches + per_round - 1) // per_round):
        for o in outs:
            o.fill_(0)
        torch.cuda.synchronize()
        with torch.cuda.stream(s2):
            for _ in range(aggr_per_round):
                aggr()
        with torch.cuda.stream(s1):
            for o in outs:
                victim.launch(o)
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
    return bad


# (id, dtype, M, N, act, ln, debug flags, launches, per_round, aggressor launches per round)
# M = 1154: two images (the 128x128 staged kernel); M = 9232: a half batch of 16 -- what one stream of the two-stream forward
# launches, served by the persistent 256x256 kernel (flag 1: its wave-private staged epilogue instead of the direct one)
_CASES = [
    ("staged-fc1-fp16", "fp16", 1154, 3072, 2, True, 0, 2400, 24, 60),
    ("staged-qkv-fp16", "fp16", 1154, 2304, 0, True, 0, 2400, 24, 60),
    ("staged-fc1-bf16", "bf16", 1154, 3072, 2, True, 0, 2400, 24, 60),
    ("direct-fc1-fp16", "fp16", 9232, 3072, 2, True, 0, 1200, 12, 160),
    ("direct-qkv-bf16", "bf16", 9232, 2304, 0, True, 0, 1200, 12, 120),
    ("direct-fc1-bf16", "bf16", 9232, 3072, 2, True, 0, 1200, 12, 160),
    ("waveprivate-fc1-fp16", "fp16", 9232, 3072, 2, True, 1, 1200, 12, 200),
    ("waveprivate-qkv-bf16", "bf16", 9232, 2304, 0, True, 1, 1200, 12, 160),
    # no fold, GELU epilogue: the remaining packed fp32 arithmetic of the GEMM translation units
    ("staged-gelu-nofold-fp16", "fp16", 1154, 3072, 2, False, 0, 2400, 24, 60),
    ("direct-gelu-nofold-bf16", "bf16", 9232, 3072, 2, False, 0, 1200, 12, 160),
]


@pytest.mark.parametrize("case", _CASES, ids=[c[0] for c in _CASES])
def test_ln_fold_consumer_bitwise_next_to_the_stem_convolution(case):
    _, dtype, M, N, act, ln, flags, launches, per_round, apr = case
    v = Victim(dtype, M, N, act=act, ln=ln, flags=flags)
    bad = victim_next_to_aggressor(v, stem_aggressor(dtype), launches, per_round, apr)
    assert bad == 0, f"{bad} of {launches} launches differ from the result computed alone (co-residency with the stem convolution)"


def _control_counts(rounds=40):
    """Child-process body: the staged fc1 / qkv victims of round 4's reproducer on whatever library DPTX_LIB names."""
    out = {}
    for name, N, act in (("fc1", 3072, 2), ("qkv", 2304, 0)):
        v = Victim("fp16", 1154, N, act=act, ln=True)
        out[name] = victim_next_to_aggressor(v, stem_aggressor("fp16", images=2), rounds * 24, 24, 60)
    return out


def test_positive_control_packed_fold_shows_the_effect():
    """The same loop on round 3's arithmetic (-DDPTX_LN_PACKED_FMA build) must find mismatches: the test can see what it guards
    against.  xfail (not fail) when the control stays clean -- the effect is a hardware behaviour, absent e.g. on another
    firmware; the product assertions above do not depend on it."""
    if not os.path.exists(CONTROL_LIB):
        pytest.skip("no control library: python tests/test_gpu_coresidency.py --build-control")
    # a control library left over from older sources lacks entry points the bindings expect (round 6: it failed the suite that
    # way once): it must carry the hash of TODAY's sources under the control's flags
    from omnidata_amd.build import source_hash
    from omnidata_amd.engine import _embedded_hash
    if _embedded_hash(CONTROL_LIB) != source_hash(CONTROL_FLAGS.split()):
        pytest.skip("stale control library (built from other sources): python tests/test_gpu_coresidency.py --build-control")
    env = dict(os.environ, DPTX_LIB=CONTROL_LIB, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--control"], capture_output=True, text=True, env=env, cwd=ROOT,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    counts = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    print("positive control (packed LayerNorm fold):", counts)
    if counts["fc1"] + counts["qkv"] == 0:
        pytest.xfail(f"the packed-fma control library did not reproduce the effect on this box: {counts}")


if __name__ == "__main__":
    if "--build-control" in sys.argv:
        # (round 6: the default build compiles the GEMM units with packed fp32 arithmetic OFF -- omnidata_amd/build.py -- so the
        #  control has to switch the target feature back on to get round 3's v_pk_fma_f32 form at all; later flags win)
        env = dict(os.environ, DPTX_CXXFLAGS=CONTROL_FLAGS, DPTX_LIB_SUFFIX="_lnpk",
                   PYTHONPATH=ROOT)
        subprocess.run([sys.executable, "-m", "omnidata_amd.build"], check=True, env=env, cwd=ROOT)
    elif "--control" in sys.argv:
        print(json.dumps(_control_counts()), flush=True)
