"""Builds omnidata_amd/libdptx.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the tree.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdptx.so")
SOURCES = ["gemm.hip", "gemm_fp16.hip", "gemm_fp16e.hip", "gemm_x3.hip", "gemm_x2.hip", "gemm_fp8.hip", "attention.hip", "norm.hip", "misc.hip", "stem.hip", "head.hip", "prepost.hip", "engine.hip"]
HEADERS = ["common.h", "kernels.h", "gemm_impl.h", os.path.join("..", "..", "include", "dptx.h")]
# Per-source compiler flags.  norm.hip / misc.hip (HBM-bound glue) are built without packed fp32 arithmetic: hipcc's SLP
# vectoriser otherwise emits v_pk_add_f32 / v_pk_fma_f32 with op_sel swizzles there (low lane reading a high dword), the
# form that misbehaved in the GEMM epilogue next to a co-resident kernel (csrc/gemm_impl.h ln_fold_fma); these
# kernels do not need the packed rate.  tests/test_build_quality.py checks every unit's ISA for that form.
SOURCE_FLAGS = {"norm.hip": ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"],
                "misc.hip": ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]}
# A/B of VERDICT r5 W11 (round 6): DPTX_GEMM_NOPK=1 builds the GEMM translation units without packed fp32 arithmetic as well
# (the suspected-erratum form then cannot be emitted there at all, whatever a toolchain bump does to the SLP vectoriser)
_NOPK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
if os.environ.get("DPTX_GEMM_NOPK") == "1":
    for _src in ("gemm.hip", "gemm_fp16.hip", "gemm_fp16e.hip", "gemm_x3.hip", "gemm_x2.hip", "gemm_fp8.hip"):
        SOURCE_FLAGS[_src] = list(_NOPK)
EXPERIMENT_HEADERS = [os.path.join("experiments", "gemm_experiments.h"), os.path.join("experiments", "gemm_experiments_dispatch.h")]


def source_hash(extra_flags=()) -> str:
    """sha256 over every source the library is built from (+ the extra compiler flags), 16 hex digits.  build() embeds it
    in the library (dptx_version() ends in `src=<hash>`) and engine.load_library() compares it with the sources next to
    the .so it is about to load: a stale binary that travelled with a newer tree is refused instead of silently used."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS + EXPERIMENT_HEADERS):
        path = os.path.join(CSRC, name)
        h.update(os.path.basename(name).encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    h.update(" ".join(extra_flags).encode())
    h.update(repr(sorted(SOURCE_FLAGS.items())).encode())
    return h.hexdigest()[:16]


def _newer(dst: str, srcs) -> bool:
    if not os.path.exists(dst):
        return False
    t = os.path.getmtime(dst)
    return all(os.path.getmtime(s) <= t for s in srcs)


def built_hash(suffix: str = "") -> str:
    """Source hash recorded by the last successful build() ("" if there was none)."""
    path = os.path.join(CSRC, "build" + suffix, "engine.srchash")
    return open(path).read().strip() if os.path.exists(path) else ""


def build(force: bool = False, verbose: bool = False) -> str:
    """DPTX_CXXFLAGS / DPTX_LIB_SUFFIX (experiments): extra compiler flags and a suffix for the object directory and the
    library name (libdptx<suffix>.so), so that kernel variants can be built side by side; engine.py loads $DPTX_LIB.
    One builder at a time per object directory (flock on <objdir>/.lock: ranks launched together by torchrun with a stale
    tree queue up, and all but the first find everything up to date); the library is linked to a temporary name and renamed
    into place, so a concurrent dlopen sees the old file or the new one, never a half-written one."""
    import fcntl
    suffix = os.environ.get("DPTX_LIB_SUFFIX", "")
    objdir = os.path.join(CSRC, "build" + suffix)
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(objdir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose, suffix, objdir)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool, suffix: str, objdir: str) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("DPTX_CXXFLAGS", "").split()
    LIB = os.path.join(HERE, f"libdptx{suffix}.so")
    hdrs = [os.path.join(CSRC, h) for h in HEADERS + EXPERIMENT_HEADERS]
    src_hash = source_hash(extra)
    hash_file = os.path.join(objdir, "engine.srchash")  # engine.o embeds the hash: rebuilt whenever any source changed
    old_hash = open(hash_file).read().strip() if os.path.exists(hash_file) else ""
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(o)
        if not force and _newer(o, [s] + hdrs) and not (src == "engine.hip" and old_hash != src_hash):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + SOURCE_FLAGS.get(src, []) + extra + ["-c", s, "-o", o]
        if src == "engine.hip":
            cmd.insert(-4, f'-DDPTX_SRC_HASH="{src_hash}"')
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    if force or procs or not _newer(LIB, objs):
        tmp = f"{LIB}.tmp{os.getpid()}"
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError(f"link failed:\n{r.stdout}")
        os.replace(tmp, LIB)
    with open(hash_file, "w") as f:
        f.write(src_hash + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
