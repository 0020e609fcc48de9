"""Round 5: several forwards in flight on one GPU (omnidata_amd/pipeline.py, include/dptx.h dptx_share_packed).  The pipeline's
handles share ONE copy of the packed weights and free-run on their own streams; every result must be the single-handle
forward's, bit for bit -- same kernels, same arithmetic, only the interleaving on the chip differs.  pytest -m gpu."""
import pytest
import torch

from omnidata_amd.engine import Engine
from omnidata_amd.model import DPTDepthModel
from omnidata_amd.pipeline import ForwardPipeline
from omnidata_amd.weights import random_dual_state_dict, random_state_dict, synthetic_input

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype", ["bf16", "mixed"])
def test_pipelined_forwards_equal_the_single_handle_forward_bitwise(dtype):
    sd = random_state_dict(0, 3)
    ref_eng = Engine(num_channels=3, max_batch=8, dtype=dtype, device_id=0)
    ref_eng.load_state_dict(sd)
    batches = [synthetic_input(20 + i, b, "normal").to(DEV) for i, b in enumerate((8, 8, 3, 8, 1, 8, 5))]   # ragged sizes too
    refs = [ref_eng.forward(x).clone() for x in batches]
    torch.cuda.synchronize()
    for depth in (2, 3):
        pipe = ForwardPipeline(depth=depth, num_channels=3, max_batch=8, dtype=dtype, device_id=0)
        pipe.load_state_dict(sd)
        for rep in range(3):   # the slots are re-used: a forward must not see its predecessor's arena content
            outs = list(pipe.map(batches))
            torch.cuda.synchronize()
            assert len(outs) == len(refs)
            for i, (o, r) in enumerate(zip(outs, refs)):
                assert torch.equal(o, r), (dtype, depth, rep, i)
        # explicit tickets into caller buffers, waited for in reverse order
        ys = [torch.empty_like(r) for r in refs]
        tickets = [pipe.submit(x, out=y) for x, y in zip(batches, ys)]
        for t in reversed(tickets):
            t.wait()
        torch.cuda.synchronize()
        assert all(torch.equal(y, r) for y, r in zip(ys, refs)) and all(t.done() for t in tickets)
        pipe.close()
    ref_eng.close()


def test_shared_weights_are_one_copy_and_unshare_on_load():
    lib = Engine(num_channels=3, max_batch=2, dtype="bf16", device_id=0)
    lib.load_state_dict(random_state_dict(0, 3))
    other = Engine(num_channels=3, max_batch=2, dtype="bf16", device_id=0)
    other.share_weights_from(lib)
    x = synthetic_input(5, 2, "normal").to(DEV)
    y0 = lib.forward(x).clone()
    assert torch.equal(other.forward(x), y0)
    # the sharer re-exports the owner's bytes
    assert torch.equal(other.export_packed(), lib.export_packed())
    # weights of its own un-share it: the owner is untouched
    other.load_state_dict(random_state_dict(1, 3))
    y1 = other.forward(x).clone()
    assert not torch.equal(y1, y0) and torch.equal(lib.forward(x), y0)
    # a handle that packs another blob is refused, with a message
    wrong = Engine(num_channels=1, max_batch=2, dtype="bf16", device_id=0)
    with pytest.raises(RuntimeError, match="share_packed"):
        wrong.share_weights_from(lib)
    wrong2 = Engine(num_channels=3, max_batch=2, dtype="fp16", device_id=0)
    with pytest.raises(RuntimeError, match="share_packed"):
        wrong2.share_weights_from(lib)
    empty = Engine(num_channels=3, max_batch=2, dtype="bf16", device_id=0)
    late = Engine(num_channels=3, max_batch=2, dtype="bf16", device_id=0)
    with pytest.raises(RuntimeError, match="share_packed"):
        late.share_weights_from(empty)    # no weights on the device yet
    # the allocation is reference-counted: the handle that loaded the weights may go first, and reloading weights into a handle
    # whose blob others read gives IT a fresh allocation (the readers keep their snapshot)
    third = Engine(num_channels=3, max_batch=2, dtype="bf16", device_id=0)
    third.share_weights_from(lib)
    lib.load_state_dict(random_state_dict(1, 3))          # lib now computes y1; third still reads the old blob
    assert torch.equal(lib.forward(x), y1) and torch.equal(third.forward(x), y0)
    fourth = Engine(num_channels=3, max_batch=2, dtype="bf16", device_id=0)
    fourth.share_weights_from(third)                        # sharing from a sharer is sharing the same allocation
    lib.close()
    third.close()
    torch.cuda.synchronize()
    assert torch.equal(fourth.forward(x), y0)
    for e in (other, wrong, wrong2, late, empty, fourth):
        e.close()


def test_dual_task_pipeline_and_model_generator():
    sd = random_dual_state_dict(0)
    eng = Engine(num_channels=3, max_batch=4, dtype="bf16", device_id=0, dual=True)
    eng.load_state_dict(sd)
    xs = [synthetic_input(40 + i, 4, "normal").to(DEV) for i in range(4)]
    refs = [tuple(t.clone() for t in eng.forward_dual(x)) for x in xs]
    pipe = ForwardPipeline.from_engine(eng, depth=2)
    outs = list(pipe.map(xs))
    torch.cuda.synchronize()
    for (n, d), (rn, rd) in zip(outs, refs):
        assert torch.equal(n, rn) and torch.equal(d, rd)
    pipe.close()
    eng.close()
    # the drop-in model's throughput form: a generator over batches, each result == model(x)
    model = DPTDepthModel(num_channels=1, dtype="mixed", max_batch=4)
    model.load_state_dict(random_state_dict(2, 1))
    model.to(DEV)
    ds = [synthetic_input(60 + i, 4, "depth").to(DEV) for i in range(5)]
    want = [model(x).clone() for x in ds]
    got = list(model.forward_pipelined(ds, depth=2))
    torch.cuda.synchronize()
    assert len(got) == 5 and all(g.shape == (4, 384, 384) for g in got)
    assert all(torch.equal(g, w) for g, w in zip(got, want))


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "mixed"])
def test_pipelined_stress_b32_bit_identical(dtype):
    """Two batch-32 forwards in flight put ViT block 0 of one forward next to the stem / first ResNetV2 stage of the other ALL the
    time -- the co-residency that exposed round 3's packed-fma fault.  120 pipelined forwards of four
    different batches, every result compared with the forward computed alone."""
    sd = random_state_dict(0, 3)
    eng = Engine(num_channels=3, max_batch=32, dtype=dtype, device_id=0, streams=1)
    eng.load_state_dict(sd)
    xs = [synthetic_input(70 + i, 32, "normal").to(DEV) for i in range(4)]
    refs = [eng.forward(x).clone() for x in xs]
    torch.cuda.synchronize()
    pipe = ForwardPipeline.from_engine(eng, depth=2)
    bad = 0
    for rnd in range(10):
        outs = list(pipe.map([xs[i % 4] for i in range(12)]))
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, refs[i % 4]) else 1 for i, o in enumerate(outs))
    pipe.close()
    eng.close()
    assert bad == 0, f"{bad} of 120 pipelined forwards differ from the forward computed alone"


def test_pipeline_from_engine_replays_layer_precision_bitwise():
    """per-layer precision set on the owner (dptx_set_layer_precision) is handle state, not part of the shared blob --
    ForwardPipeline.from_engine must replay it, or the pipelined results are the DEFAULT table's (and bench.py's
    parity_mode_head0_2mfma line would time one table and check another)."""
    sd = random_state_dict(0, 3)
    own = Engine(num_channels=3, max_batch=4, dtype="mixed", device_id=0)
    own.load_state_dict(sd)
    x = synthetic_input(31, 4, "normal").to(DEV)
    y_default = own.forward(x).clone()
    own.set_layer_precision("scratch.output_conv.0.weight", 2)
    y_two = own.forward(x).clone()
    torch.cuda.synchronize()
    assert not torch.equal(y_default, y_two)          # the override changes the arithmetic ...
    pipe = ForwardPipeline.from_engine(own, depth=2)
    assert all(e.layer_precision == {"scratch.output_conv.0.weight": 2} for e in pipe.engines)
    outs = list(pipe.map([x, x, x]))
    torch.cuda.synchronize()
    assert all(torch.equal(o, y_two) for o in outs)   # ... and the pipeline's handles compute what the owner computes
    pipe.close()
    own.close()


def test_rejected_import_leaves_the_handle_as_it_was():
    """dptx_import_packed_device validates the incoming blob's layout header BEFORE it un-shares / re-allocates: a
    handle that shares another's weights keeps them (and keeps working) when the import is refused."""
    lib = Engine(num_channels=3, max_batch=2, dtype="bf16", device_id=0)
    lib.load_state_dict(random_state_dict(0, 3))
    sharer = Engine(num_channels=3, max_batch=2, dtype="bf16", device_id=0)
    sharer.share_weights_from(lib)
    x = synthetic_input(6, 2, "normal").to(DEV)
    y0 = lib.forward(x).clone()
    bad = lib.export_packed().clone()
    bad[:8] = 0   # not a DPTXBLOB header
    with pytest.raises(RuntimeError, match="another layout"):
        sharer.import_packed(bad)
    assert torch.equal(sharer.forward(x), y0) and torch.equal(lib.forward(x), y0)
    assert torch.equal(sharer.export_packed(), lib.export_packed())   # still the shared blob
    with pytest.raises(RuntimeError, match="another layout"):
        lib.import_packed(bad)                                        # the owner, whose blob somebody else reads
    assert torch.equal(lib.forward(x), y0) and torch.equal(sharer.forward(x), y0)
    sharer.close()
    lib.close()


def test_model_forward_pipelined_caches_chunks_and_matches_model():
    """DPTDepthModel.forward_pipelined: one cached pipeline per engine, batches above max_batch chunked like
    model(x), results bit-identical to model(x)."""
    model = DPTDepthModel(num_channels=3, dtype="mixed", max_batch=4, init_seed=0).eval()
    xs = [synthetic_input(40 + i, b, "normal").to(DEV) for i, b in enumerate((4, 9, 2))]   # 9 > max_batch
    refs = [model(x).clone() for x in xs]
    got = list(model.forward_pipelined(xs, depth=2))
    pipe = model._pipe
    assert pipe is not None and len(got) == 3 and all(torch.equal(g, r) for g, r in zip(got, refs))
    got2 = list(model.forward_pipelined(xs[:2], depth=2))
    assert model._pipe is pipe and all(torch.equal(g, r) for g, r in zip(got2, refs))    # same pipeline, no re-allocation
    model.load_state_dict(random_state_dict(1, 3))                                       # new weights: the engine and its pipeline go
    got3 = list(model.forward_pipelined(xs[:1], depth=2))
    assert model._pipe is not pipe and not torch.equal(got3[0], refs[0]) and torch.equal(got3[0], model(xs[0]))


def test_stream_overlap_probe_tells_concurrent_from_serialised():
    """nothing used to check that the pipeline's streams overlap.  dptx_probe_stream_overlap on ONE stream twice
    (the forced-serial case: two launches on one queue) must read ~2, on the pipeline's two slot streams ~1; ForwardPipeline
    records the ratio it found and warns when it cannot get below 1.5."""
    from omnidata_amd.engine import probe_stream_overlap
    s = torch.cuda.Stream()
    serial = probe_stream_overlap(s, s, 0)
    assert 1.7 < serial < 2.4, serial                      # same stream = same queue: one after the other
    pipe = ForwardPipeline(depth=2, num_channels=3, max_batch=2, dtype="bf16", device_id=0)
    print(f"\nprobe: one stream twice {serial:.2f}, pipeline slots {pipe.overlap_ratio:.2f}")
    assert pipe.overlap_ratio is not None and pipe.overlap_ratio < ForwardPipeline.SERIALISED
    assert abs(pipe.measure_overlap() - pipe.overlap_ratio) < 0.4
    # a pipeline whose two slots are the SAME stream serialises, and says so
    import warnings
    pipe.streams = [s, s]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        pipe._levels = [0]
        r = ForwardPipeline.ensure_overlap(pipe, retries=0)
    assert r > 1.7 and any("do not run concurrently" in str(m.message) for m in w)
    pipe.close()
