"""Static checks on the compiled gfx950 code objects (hipcc cross-compiles here, no GPU needed):
no kernel may use scratch memory (a staging array that falls out of registers serialises every
global load behind s_waitcnt vmcnt(0) -- this happened once) and the hot kernels must contain the
instructions the design relies on."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "omnidata_amd", "csrc")


ALL = ["gemm.hip", "gemm_fp16.hip", "gemm_fp16e.hip", "gemm_x3.hip", "gemm_x2.hip", "gemm_fp8.hip", "attention.hip", "norm.hip", "misc.hip", "stem.hip", "head.hip",
       "prepost.hip"]
_cache = {}


def disasm(src, tmp_path):
    """Device assembly of one translation unit; the first call compiles ALL of them side by side (the four GEMM units take
    about a minute each)."""
    if not _cache:
        procs = {}
        for name in ALL:
            out = tmp_path / (name + ".s")
            from omnidata_amd.build import SOURCE_FLAGS
            procs[name] = (out, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17"] +
                                                 SOURCE_FLAGS.get(name, []) + ["-S", "--cuda-device-only", "-o", str(out),
                                                                               os.path.join(CSRC, name)],
                                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE))
        for name, (out, pr) in procs.items():
            _, err = pr.communicate()
            assert pr.returncode == 0, err.decode()[-2000:]
            _cache[name] = out.read_text()
    return _cache[src]


@pytest.mark.parametrize("src", ALL)
def test_no_scratch_no_spills(src, tmp_path):
    s = disasm(src, tmp_path)
    names = re.findall(r"^\s+\.name:\s+(\S+)", s, flags=re.M)
    priv = re.findall(r"^\s+\.private_segment_fixed_size:\s+(\d+)", s, flags=re.M)
    spills = re.findall(r"^\s+\.vgpr_spill_count:\s+(\d+)", s, flags=re.M)
    assert names and len(priv) >= 1
    assert all(int(p) == 0 for p in priv), dict(zip(names, priv))
    assert all(int(p) == 0 for p in spills)
    assert "scratch_" not in s
    if src.startswith("gemm"):  # gemm_impl.h instantiated per translation unit
        want = {"gemm.hip": ["v_mfma_f32_32x32x16_bf16"], "gemm_fp16.hip": ["v_mfma_f32_32x32x16_f16"],
                "gemm_fp16e.hip": ["v_mfma_f32_32x32x16_f16"],
                "gemm_x3.hip": ["v_mfma_f32_32x32x16_bf16", "v_mfma_f32_32x32x16_f16"],
                "gemm_x2.hip": ["v_mfma_f32_32x32x16_f16"],
                "gemm_fp8.hip": ["v_mfma_scale_f32_32x32x64_f8f6f4"]}[src]
        assert all(w in s for w in want)
        assert re.search(r"buffer_load_dwordx4 .* lds", s), "direct-to-LDS staging missing"
        if src in ("gemm.hip", "gemm_fp16.hip", "gemm_fp16e.hip"):
            assert "v_pk_max_i16" in s  # packed ReLU on the A fragments
    if src == "attention.hip":
        assert "v_mfma_f32_32x32x16_bf16" in s and "v_exp_f32" in s


@pytest.mark.parametrize("src", ALL)
def test_no_packed_fp32_op_reads_a_high_dword_in_its_low_lane(src, tmp_path):
    """Round 4 root cause of the 'nondeterministic' multi-stream forward: v_pk_fma_f32 ... op_sel:[0,1,0] (hipcc's packed form
    of the LayerNorm-fold epilogue: both lanes take rstd from the HIGH dword of a register pair) returned a product of zero in
    the low lane for work-items 48..63 whenever the stem convolution of another stream shared the CU (tools/gpu/r4_micro.py:
    224-445 of 2400 launches; the scalar form: 0).  No kernel of the library may contain a packed fp32 operation with an
    op_sel swizzle on the low lane (op_sel_hi -- the HIGH lane reading a low dword, what a splat compiles to -- is what
    the GELU polynomial uses and never showed the effect)."""
    s = disasm(src, tmp_path)
    bad = [ln.strip() for ln in s.splitlines() if re.search(r"\bv_pk_\w+_f32\b", ln) and re.search(r"op_sel:\[[01,]*1", ln)]
    assert not bad, f"{len(bad)} packed fp32 ops with a low-lane op_sel swizzle, e.g. {bad[:3]}"
