import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


# Collection order of the -m gpu suite: evidence against the ORACLE first (kernel-level op tests, then the 1e-3 modes,
# then end-to-end / golden / other sizes / pre-post-processing), comparisons of the engine with ITSELF (determinism,
# fused-vs-unfused, stream schedules, dual-vs-single, CLI plumbing) last -- with `-x`, a failing self-comparison can no
# longer hide the parity evidence.
_FILE_RANK = ["test_gpu_ops.py", "test_gpu_x3.py", "test_gpu_mixed.py", "test_gpu_e2e.py", "test_gpu_vitl16.py", "test_gpu_flex.py",
              "test_gpu_prepost.py", "test_gpu_fp8.py", "test_gpu_dual.py", "test_gpu_cli.py", "test_gpu_poison.py", "test_gpu_stress.py"]
_SELF_COMPARISONS = ("deterministic", "fused_head_equals", "two_stream", "packed_blob", "bitwise", "bit_identical", "stress",
                     "input_contract", "prior_arena")


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        fname = os.path.basename(str(item.fspath))
        rank = _FILE_RANK.index(fname) if fname in _FILE_RANK else len(_FILE_RANK)
        is_self = any(t in item.name for t in _SELF_COMPARISONS)
        return (1 if is_self else 0, rank)

    items.sort(key=key)  # stable: the order inside a file is kept


@pytest.fixture(scope="session")
def built_lib():
    """libdptx.so, built in-tree if missing (hipcc cross-compiles without a GPU)."""
    from omnidata_amd.build import build
    return build()


@pytest.fixture(autouse=True)
def _seed_torch_rngs(request):
    """Every test starts from the same torch RNG state, host and device.  Several GPU tests draw small operands (biases, head
    weights) with torch.randn(device=...) and no generator: their error-vs-fp64 margins then depended on what ran before them
    (round 6: test_fp16x3_fused_head_tail[1-192-192-1-1] read 2.10e-5 against a 2e-5 bound once in six full runs)."""
    import torch
    torch.manual_seed(20260930)   # seeds the CUDA generators too (lazily: nothing happens on a box without a GPU)
    yield
