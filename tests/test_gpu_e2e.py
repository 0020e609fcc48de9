"""End-to-end parity of the HIP engine (through the C ABI) against the CPU fp32 oracle and the
committed golden vectors.  Run on an MI355X: pytest -m gpu.

Tolerances.  north_star asks for 1e-3 abs against the PyTorch-CPU fp32 forward.  Rounding the
two operands of every MFMA to 16 bit (fp32 accumulate, fp32 everything else) already costs,
on these seeded weights, max 5.5e-2 / rms 1.1e-2 in bf16 and max 7.2e-3 / rms 1.5e-3 in fp16
(oracle/precision_study.py, emulated on CPU) -- that is a property of single-pass 16-bit
arithmetic, not of this engine, and it applies equally to the reference under autocast.  The
tests below therefore pin (a) the implementation tightly at op and stage level and (b) the
end-to-end deviation at ~1.3x the emulated floor; DESIGN.md "Parity" states the gap to 1e-3.
"""
import glob
import os

import numpy as np
import pytest
import torch

from omnidata_amd.model import DPTDepthModel
from omnidata_amd.weights import random_state_dict, synthetic_input
from oracle.dpt_oracle import dpt_forward, mean_angular_error_deg, oracle_threads, ssi_align
from oracle.validate_vs_reference import subsample

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# (max-abs, rms) budget on the final [0,1]-range output vs the fp32 oracle
E2E_TOL = {"bf16": (8e-2, 1.6e-2), "fp16": (1.2e-2, 2.5e-3)}
# rms error / rms value allowed at every stage tap.  The emulated floor (operand rounding only,
# oracle/precision_study.py:stage_errors) peaks at ResNet stage 2 with these synthetic weights:
# bf16 1.15e-1, fp16 1.6e-2; the engine additionally rounds conv outputs before GroupNorm and
# measured 1.30e-1 / 1.9e-2.  Budget = ~1.5x the emulated floor.
STAGE_RMS_REL = {"bf16": 1.7e-1, "fp16": 2.5e-2}
_cache, _oracle = {}, {}


def oracle_case(task, C, seed, B):
    key = (task, seed, B)
    if key not in _oracle:
        oracle_threads()
        sd = random_state_dict(seed, C)
        x = synthetic_input(seed, B, task)
        taps = {}
        _oracle[key] = (sd, x, dpt_forward(sd, x, taps), taps)
    return _oracle[key]


def run_case(task, C, seed, B, dtype, taps=False):
    key = (task, seed, B, dtype, taps)
    if key in _cache:
        return _cache[key]
    sd, x, ref, otaps = oracle_case(task, C, seed, B)
    model = DPTDepthModel(num_channels=C, dtype=dtype, max_batch=max(B, 1))
    model.load_state_dict(sd)
    model.to(DEV)
    if taps:
        model._get_engine(torch.device(DEV)).enable_taps(True)
    y = model(x.to(DEV)).cpu()
    _cache[key] = (y, ref, model, otaps)
    return _cache[key]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("task,C,seed,B", [("normal", 3, 0, 1), ("depth", 1, 0, 1), ("normal", 3, 1, 2)])
def test_engine_vs_oracle(task, C, seed, B, dtype):
    y, ref, _, _ = run_case(task, C, seed, B, dtype)
    assert y.shape == ref.shape  # [B,3,384,384] or squeezed [B,384,384]
    assert torch.isfinite(y).all() and (y >= 0).all()
    d = (y - ref).abs()
    mx, rms = d.max().item(), d.pow(2).mean().sqrt().item()
    print(f"\n[{task} seed={seed} B={B} {dtype}] max|d|={mx:.3e} rms={rms:.3e} out std={ref.std():.3f}")
    assert mx < E2E_TOL[dtype][0] and rms < E2E_TOL[dtype][1]
    if task == "normal":
        ang = mean_angular_error_deg(y.clamp(0, 1), ref.clamp(0, 1))
        print(f"    mean angular error {ang:.3f} deg")
        assert ang < (6.0 if dtype == "bf16" else 1.0)
    else:
        ds = (ssi_align(y, ref) - ref).abs().max().item()
        print(f"    scale/shift-aligned max|d|={ds:.3e}")
        assert ds < E2E_TOL[dtype][0]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_stage_taps_track_oracle(dtype):
    """Every stage boundary against the oracle's tap: catches a wrong layer even
    when later normalisation would hide it in the final output."""
    y, ref, model, otaps = run_case("normal", 3, 0, 1, dtype, taps=True)
    eng = model.engine
    names = ["stem", "s0", "s1", "s2", "tok0", "blk0", "blk3", "blk8", "blk11", "l3", "l4", "l1_rn", "l2_rn",
             "l3_rn", "l4_rn", "p4", "p3", "p2", "p1", "h0", "h1"]
    worst, bad = 0.0, []
    for n in names:
        got, want = eng.tap(n), otaps[n]
        assert got.shape == want.shape, (n, got.shape, want.shape)
        rel = ((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
        print(f"    tap {n:6s} rms-rel err {rel:.3e}")
        worst = max(worst, rel)
        if not rel < STAGE_RMS_REL[dtype]:
            bad.append((n, rel))
    print(f"[{dtype}] worst stage rms-rel error {worst:.3e}")
    assert not bad, bad


GOLDEN_TOL = {**E2E_TOL, "bf16x3": (1e-3, 2e-4)}   # bf16x3 is held to north_star's 1e-3 against the reference's own output


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "bf16x3"])
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "dpt_*.npz"))),
                         ids=lambda p: os.path.basename(p))
def test_engine_vs_reference_golden(path, dtype):
    """Against vectors produced by the reference's OWN modules (tests/golden/), in the benchmarked dtype (bf16), fp16 and
    the parity mode (bf16x3, 1e-3)."""
    g = np.load(path)
    task, C, seed, B = str(g["task"]), int(g["num_channels"]), int(g["seed"]), int(g["batch"])
    y, _, _, _ = run_case(task, C, seed, B, dtype)
    tol = GOLDEN_TOL[dtype]
    d = np.abs(subsample(y) - g["out_sub"])
    print(f"\n[{os.path.basename(path)} {dtype}] vs reference golden: max|d|={d.max():.3e} rms={np.sqrt((d ** 2).mean()):.3e}")
    assert d.max() < tol[0] and np.sqrt((d ** 2).mean()) < tol[1]
    row = y.reshape(B, -1, 384, 384)[0, 0, 191].numpy()
    assert np.abs(row - g["out_row"]).max() < tol[0]


@pytest.mark.parametrize("io", [torch.bfloat16, torch.float16])
def test_sixteen_bit_caller_io(io):
    """DPTX_IO_BF16 / _FP16 (SURVEY.md 8d config 2 feeds bf16): the stem reads and the head writes 16-bit NCHW tensors.
    Feeding the 16-bit image must equal feeding its fp32 copy bit for bit before the output rounding, i.e. the 16-bit
    result is exactly the rounding of the fp32 result -- and it is held to the same budget against the oracle."""
    sd = random_state_dict(0, 3)
    model = DPTDepthModel(num_channels=3, dtype="bf16", max_batch=3)
    model.load_state_dict(sd)
    model.to(DEV)
    x16 = synthetic_input(2, 3, "normal").to(io)
    y16 = model(x16.to(DEV))
    assert y16.dtype == io and y16.shape == (3, 3, 384, 384)
    y32 = model(x16.float().to(DEV))
    assert y32.dtype == torch.float32
    assert torch.equal(y16, y32.to(io))
    oracle_threads()
    ref = dpt_forward(sd, x16.float())
    d = (y32.cpu() - ref).abs()
    assert d.max() < E2E_TOL["bf16"][0] and d.pow(2).mean().sqrt() < E2E_TOL["bf16"][1]


def test_deterministic_and_batch_invariant():
    """Size-independent properties at the benchmark batch: two runs are bit-identical (no atomics),
    and image i of a batch of 32 equals the same image run alone."""
    sd = random_state_dict(0, 3)
    model = DPTDepthModel(num_channels=3, dtype="bf16", max_batch=32)
    model.load_state_dict(sd)
    model.to(DEV)
    x = synthetic_input(5, 32, "normal").to(DEV)
    y1 = model(x).clone()
    y2 = model(x)
    assert torch.equal(y1, y2)
    for i in (0, 17, 31):
        yi = model(x[i:i + 1])
        assert torch.equal(yi[0], y1[i]), i
    # chunking over max_batch
    small = DPTDepthModel(num_channels=3, dtype="bf16", max_batch=3)
    small.load_state_dict(sd)
    small.to(DEV)
    assert torch.equal(small(x[:7]), y1[:7])


def test_input_contract_errors():
    model = DPTDepthModel(num_channels=1, dtype="bf16", max_batch=1).to(DEV)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.zeros(1, 3, 384, 384))
    with pytest.raises(ValueError):
        model(torch.zeros(1, 3, 250, 256, device=DEV))  # not a multiple of 32 (tests/test_gpu_flex.py covers other sizes)
    y = model(torch.zeros(1, 3, 384, 384, device=DEV))
    assert y.shape == (1, 384, 384)
    n, alg, exe = model.engine.info()
    assert 150 < n < 300 and abs(alg - 127.615e9) < 1e6 and 120e9 < exe < 130e9  # 199 launches (225 without the LayerNorm fold)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_fused_head_equals_unfused(dtype):
    """The one-launch head tail (head.hip) against the three-launch path the engine takes while stage taps are on.  The
    two differ only in h1, the 32-channel map in front of the 1x1 projection: fp32 registers in the fused kernel, rounded
    to 16 bit in memory otherwise (plus fp32 summation order, ~1e-6).  So per output pixel
        |y_fused - y_unfused| <= sum_c |w4[o,c]| * (half an ulp of h1[c] at that pixel)
    which is evaluated from the recorded h1 tap -- a bound derived from the data, not a typed-in constant -- and both
    paths are held to the same budget against the fp32 oracle."""
    sd = random_state_dict(2, 3)
    model = DPTDepthModel(num_channels=3, dtype=dtype, max_batch=2)
    model.load_state_dict(sd)
    model.to(DEV)
    x = synthetic_input(3, 2, "normal")
    xg = x.to(DEV)
    y_fused = model(xg).clone()
    eng = model.engine
    launches_fused = eng.info()[0]  # one run over the whole batch (streams = 0: one stream until a schedule has been measured)
    eng.enable_taps(True)
    y_unfused = model(xg).clone()      # taps: three launches instead of one for the head tail
    assert launches_fused == eng.info()[0] - 2
    h1 = eng.tap("h1").double()        # [B,32,384,384], the 16-bit values the unfused 1x1 projection read
    eng.enable_taps(False)
    mant = 8 if dtype == "bf16" else 11   # significand bits
    half_ulp = torch.where(h1 > 0, torch.exp2(torch.floor(torch.log2(h1.clamp_min(1e-30))) - mant), torch.zeros_like(h1))
    w4 = sd["scratch.output_conv.4.weight"].double().abs().reshape(3, 32)
    # 1.02: a value within fp32 noise of a rounding boundary may round the other way; 2e-6: fp32 summation order
    bound = (1.02 * torch.einsum("oc,bchw->bohw", w4, half_ulp) + 2e-6).float()
    d = (y_fused - y_unfused).abs().cpu()
    print(f"\n[{dtype}] fused vs unfused head tail: max|d| = {d.max():.3e}, max bound {bound.max():.3e}, "
          f"worst d/bound {(d / bound).max():.3f}")
    assert (d <= bound).all()
    oracle_threads()
    ref = dpt_forward(sd, x)
    for name, y in (("fused", y_fused), ("unfused", y_unfused)):
        e = (y.cpu() - ref).abs()
        assert e.max() < E2E_TOL[dtype][0] and e.pow(2).mean().sqrt() < E2E_TOL[dtype][1], name
    assert torch.equal(model(xg), y_fused)


def test_two_stream_split_is_bit_identical():
    """dptx_config.streams: two half-batches on two internal streams (default) against everything on the caller's stream."""
    from omnidata_amd.engine import Engine
    sd = random_state_dict(0, 3)
    x = synthetic_input(5, 7, "normal").to(DEV)
    outs = []
    for streams in (1, 2):
        eng = Engine(num_channels=3, max_batch=7, dtype="bf16", device_id=0, streams=streams)
        eng.load_state_dict(sd)
        y = eng.forward(x).clone()
        for _ in range(12):          # sporadic cross-stream hazards show up as single-image differences in some runs
            assert torch.equal(eng.forward(x), y)
        side = torch.cuda.Stream()   # and from a non-default caller stream
        with torch.cuda.stream(side):
            side.wait_stream(torch.cuda.current_stream())
            y2 = eng.forward(x[:5]).clone()
        side.synchronize()
        assert torch.equal(y2, y[:5])
        outs.append(y)
        n, _, _ = eng.info()
        assert (n > 300) == (streams == 2)   # 2 x 199 launches on two streams, 199 on one
        eng.close()
    assert torch.equal(outs[0], outs[1])


def test_measured_schedule_choice_is_bit_identical_and_reported():
    """dptx_config.streams = 0 (the default, round 6): one stream until dptx_tune_schedule has timed both intra-forward schedules
    on the caller's batch; the two-half schedule is kept only if it was >= 3 % faster; either way the bits are the same.
    Engine.forward tunes by itself at the first batch of >= 8 images; explicit streams are never overridden."""
    from omnidata_amd.engine import Engine
    sd = random_state_dict(0, 3)
    x = synthetic_input(9, 8, "normal").to(DEV)
    ref = Engine(num_channels=3, max_batch=8, dtype="bf16", device_id=0, streams=1)
    ref.load_state_dict(sd)
    y_ref = ref.forward(x).clone()
    assert ref.schedule_info() == {"split": False, "tuned": False, "ms_single": 0.0, "ms_split": 0.0}
    eng = Engine(num_channels=3, max_batch=8, dtype="bf16", device_id=0)     # streams = 0
    eng.share_weights_from(ref)
    assert eng.schedule_info()["tuned"] is False and eng.schedule_info()["split"] is False
    y_small = eng.forward(x[:3]).clone()                                      # below AUTO_TUNE_MIN_BATCH: no measurement
    assert eng.schedule_info()["tuned"] is False and torch.equal(y_small, y_ref[:3]) and eng.info()[0] < 300
    y = eng.forward(x).clone()                                                # first big batch: measured, then run
    info = eng.schedule_info()
    print(f"\nmeasured schedule: {info}")
    assert info["tuned"] and info["ms_single"] > 0 and info["ms_split"] > 0
    assert info["split"] == (info["ms_split"] < 0.97 * info["ms_single"])
    assert torch.equal(y, y_ref)
    for _ in range(4):
        assert torch.equal(eng.forward(x), y_ref)
    assert (eng.info()[0] > 300) == info["split"]                             # the chosen schedule is the one that runs
    two = Engine(num_channels=3, max_batch=8, dtype="bf16", device_id=0, streams=2)   # explicit: never overridden
    two.share_weights_from(ref)
    assert torch.equal(two.forward(x), y_ref) and two.info()[0] > 300
    assert two.schedule_info()["split"] is True and two.schedule_info()["tuned"] is False
    for e in (two, eng, ref):
        e.close()


def test_packed_blob_cache_roundtrip(tmp_path):
    """save_packed on one engine, load_packed into a fresh one: same forward, no fp32 weights needed."""
    from omnidata_amd.engine import Engine
    sd = random_state_dict(4, 1)
    a = Engine(num_channels=1, max_batch=2, dtype="bf16", device_id=0)
    a.load_state_dict(sd)
    path = str(tmp_path / "depth.safetensors")
    a.save_packed(path)
    b = Engine(num_channels=1, max_batch=2, dtype="bf16", device_id=0)
    b.load_packed(path)
    x = synthetic_input(1, 2, "depth").to(DEV)
    assert torch.equal(a.forward(x), b.forward(x))


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "mixed"])
def test_layernorm_fold_matches_separate_layernorm(dtype):
    """The LayerNorm of the ViT blocks is folded into the qkv / fc1 GEMMs (include/dptx.h DPTX_FLAG_NO_LN_FOLD; the row
    statistics and the 16-bit operand copy of the fp32 token stream come out of the preceding proj / fc2 / patch-embed
    epilogue).  Both schedules are the same arithmetic up to WHERE the 16-bit rounding of the qkv / fc1 operand happens
    (before vs after the normalisation), so they must agree far inside the dtype's budget against the oracle, every token
    tap included -- a wrong mean / rstd / column sum shows up at blk0 at once."""
    from omnidata_amd.engine import Engine
    sd, x, ref, otaps = oracle_case("normal", 3, 0, 2)
    taps, outs = {}, {}
    # flags: 0 = default (fold; the single-pass dtypes also keep the token stream in 16 bit only), 4 = fold with the fp32
    # stream (DPTX_FLAG_FP32_STREAM), 1 = separate LayerNorm launches (DPTX_FLAG_NO_LN_FOLD)
    for flags in (0, 4, 1):
        eng = Engine(num_channels=3, max_batch=2, dtype=dtype, device_id=0, flags=flags)
        eng.load_state_dict(sd)
        eng.enable_taps(True)
        outs[flags] = eng.forward(x.to(DEV)).cpu()
        taps[flags] = {n: eng.tap(n) for n in ("tok0", "blk0", "blk5", "blk11", "l3", "l4")}
        eng.close()
    step = {"bf16": 2.0 ** -8, "fp16": 2.0 ** -11, "mixed": 2.0 ** -11}[dtype]
    assert torch.equal(taps[4]["tok0"], taps[1]["tok0"])   # the fp32 stream itself is written identically
    if dtype == "mixed":
        assert torch.equal(taps[0]["tok0"], taps[1]["tok0"])   # the parity mode always keeps the fp32 stream
    else:  # 16-bit stream: tok0 is the 16-bit rounding of the same values
        assert (taps[0]["tok0"] - taps[1]["tok0"]).abs().max() <= step * taps[1]["tok0"].abs().max()
    for n in ("blk0", "blk5", "blk11", "l3", "l4"):
        want = otaps[n]
        rel = {f: ((taps[f][n] - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item() for f in (0, 4, 1)}
        rel_ab = ((taps[0][n] - taps[1][n]).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
        print(f"    [{dtype}] tap {n:6s} default vs separate {rel_ab:.3e}; vs oracle: default {rel[0]:.3e} fold+fp32 stream {rel[4]:.3e} "
              f"separate {rel[1]:.3e}")
        assert rel_ab < 12 * step, n                 # a handful of operand roundings apart
        assert rel[0] < 3 * rel[1] + 4 * step, n      # and neither fused schedule is markedly less accurate
        assert rel[4] < 3 * rel[1] + 4 * step, n
    d0, d1 = (outs[0] - ref).abs().max().item(), (outs[1] - ref).abs().max().item()
    print(f"    [{dtype}] out max|d| vs oracle: fold {d0:.3e} separate {d1:.3e}")
    bar = {"bf16": E2E_TOL["bf16"][0], "fp16": E2E_TOL["fp16"][0], "mixed": 1e-3}[dtype]
    assert d0 < bar


def test_gemm_launch_forms_agree_bit_for_bit():
    """The 256x256 GEMM kernel has three launch forms -- register-direct epilogue on transposed accumulators, staged
    epilogue, one block per tile instead of the persistent tile loop (dptx_debug_set_gemm_flags) -- and a layer's form depends on
    its shape, i.e. on the batch size: all of them must produce the same bits (explicit fmas in every epilogue; the MFMA is
    symmetric under operand swap, tools/gpu/probes/mfma_swap.hip).  Whole forward at B = 32, where the ViT GEMMs, the RCU
    convolutions and the head take that kernel, in bf16 and in the parity mode."""
    from omnidata_amd.engine import load_library
    lib = load_library()
    x = synthetic_input(9, 32, "normal").to(DEV)
    try:
        for dtype in ("bf16", "mixed"):
            model = DPTDepthModel(num_channels=3, dtype=dtype, max_batch=32)
            model.load_state_dict(random_state_dict(0, 3))
            model.to(DEV)
            outs = []
            for flags in (0, 1, 3):
                lib.dptx_debug_set_gemm_flags(flags)
                outs.append(model(x).clone())
            assert torch.equal(outs[0], outs[1]), dtype   # direct == staged
            assert torch.equal(outs[1], outs[2]), dtype   # persistent == one block per tile
    finally:
        lib.dptx_debug_set_gemm_flags(0)
