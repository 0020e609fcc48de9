"""Long-running bit-identity stress of the multi-stream schedule (VERDICT r1 W4 / ADVICE): the forward of >= 2 images runs
as sub-batches on internal HIP streams; every producer -> consumer edge inside a sub-batch is a kernel boundary on one
stream while the other streams keep the GPU busy.  tools/gpu/stale_probe.hip showed that such a boundary orders execution
and memory under concurrency (0 stale reads in 120 000 iterations); this test holds the engine itself to it:
1000 forwards on two and on three streams, each compared bit for bit with the single-stream result.  pytest -m gpu."""
import pytest
import torch

from omnidata_amd.engine import Engine
from omnidata_amd.weights import random_state_dict, synthetic_input

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype,streams,iters", [("bf16", 2, 1000), ("bf16", 3, 300), ("mixed", 2, 400)])
def test_multi_stream_schedule_stress_bit_identical(dtype, streams, iters):
    """bf16: the benched throughput mode; mixed: the parity mode (two-plane arena, 3-MFMA tiles, the default dtype of the
    drop-in surface)."""
    sd = random_state_dict(0, 3)
    B = 6
    x = synthetic_input(5, B, "normal").to(DEV)
    ref_eng = Engine(num_channels=3, max_batch=B, dtype=dtype, device_id=0, streams=1)
    ref_eng.load_state_dict(sd)
    ref = ref_eng.forward(x).clone()
    ref_eng.close()
    eng = Engine(num_channels=3, max_batch=B, dtype=dtype, device_id=0, streams=streams)
    eng.load_state_dict(sd)
    out = torch.empty_like(ref)
    # every forward is checked on the device without stalling the streams (the comparison kernels queue up behind the
    # join event on the caller's stream while the next forward's sub-batches already run on the internal streams)
    mism = torch.zeros((), dtype=torch.int64, device=DEV)
    for _ in range(iters):
        eng.forward(x, out=out)
        mism += (out != ref).sum()
    assert int(mism.item()) == 0
    eng.close()
