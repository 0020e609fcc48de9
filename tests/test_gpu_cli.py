"""Drop-in surface on the GPU: the demo.py CLI (BASELINE configs[0] plumbing, but through the HIP path), torch.hub
entry points, batch sizes around max_batch, and the N>1 code path of bench.py (2 ranks sharing the one GPU over
gloo: packed-weight broadcast -> import -> independent batches -> max-over-ranks timing).  pytest -m gpu."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **kw):
    env = dict(os.environ, PYTHONPATH=ROOT)
    return subprocess.run([sys.executable] + args, capture_output=True, text=True, env=env, cwd=kw.get("cwd", ROOT), timeout=600)


def test_demo_cli_writes_reference_file_names(tmp_path):
    rng = np.random.default_rng(0)
    src = tmp_path / "in"
    src.mkdir()
    Image.fromarray(rng.integers(0, 255, (512, 640, 3), dtype=np.uint8)).save(src / "test1.png")
    Image.fromarray(rng.integers(0, 255, (400, 400), dtype=np.uint8)).save(src / "grey.png")   # 1-channel input
    out = tmp_path / "out"
    for task in ("normal", "depth"):
        r = _run([os.path.join(ROOT, "demo.py"), "--task", task, "--img_path", str(src), "--output_path", str(out),
                  "--random-weights", "0"])
        assert r.returncode == 0, r.stderr[-2000:]
        for stem in ("test1", "grey"):
            assert f"Writing output {out}/{stem}_{task}.png" in r.stdout.replace(os.sep, "/")
            pred = Image.open(out / f"{stem}_{task}.png")
            assert pred.size == ((384, 384) if task == "normal" else (512, 512))      # demo.py:150 / :143
            assert Image.open(out / f"{stem}_rgb.png").size == (512, 512)            # demo.py:101-102,134
    # single-file form
    r = _run([os.path.join(ROOT, "demo.py"), "--task", "normal", "--img_path", str(src / "test1.png"), "--output_path",
              str(tmp_path / "o2"), "--random-weights", "1", "--dtype", "fp16"])
    assert r.returncode == 0 and (tmp_path / "o2" / "test1_normal.png").exists()
    # DPT-Large (demo.py:81) in the parity mode
    r = _run([os.path.join(ROOT, "demo.py"), "--task", "depth", "--img_path", str(src / "test1.png"), "--output_path",
              str(tmp_path / "o3"), "--random-weights", "2", "--backbone", "vitl16_384", "--dtype", "mixed"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert Image.open(tmp_path / "o3" / "test1_depth.png").size == (512, 512)


def test_hub_model_and_batch_sizes():
    from omnidata_amd.weights import synthetic_input
    m = torch.hub.load(ROOT, "dpt_hybrid_384", source="local", pretrained=False, task="normal", max_batch=4).to("cuda:0")
    x = synthetic_input(3, 9, "normal").to("cuda:0")
    y9 = m(x)                      # 9 > max_batch: chunks of 4,4,1
    assert y9.shape == (9, 3, 384, 384) and torch.isfinite(y9).all()
    for b in (1, 2, 3, 4, 5):
        assert torch.equal(m(x[:b]), y9[:b])
    d = torch.hub.load(ROOT, "depth_dpt_hybrid_384", source="local", pretrained=False).to("cuda:0")
    assert d(synthetic_input(0, 2, "depth").to("cuda:0")).shape == (2, 384, 384)
    # weights can be replaced after the engine exists (state_dict round trip)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    sd["scratch.output_conv.4.bias"] += 0.25
    m.load_state_dict(sd)
    y2 = m(x[:1])
    assert torch.allclose(y2, y9[:1] + 0.25, atol=1e-5) or bool((y2 - y9[:1]).abs().max() > 0.2)


def test_bench_two_ranks_share_one_gpu_over_gloo():
    r = _run(["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", "29577", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
              "--batch", "4", "--dist-backend", "gloo", "--share-gpu", "--no-cpu-baseline", "--profile-steps", "1"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 8 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["roofline"]["frac"] > 0


def test_bench_plain_python_gpus_2_spawns_two_ranks():
    """the plain `python bench.py --gpus 2` form (no launcher) starts the two ranks itself; the line proves it."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--dist-backend", "gloo", "--share-gpu", "--no-cpu-baseline", "--profile-steps", "1"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and cfg["ranks_seen"] == 2 and cfg["global_batch"] == 8 and len(cfg["rank_devices"]) == 2
    assert cfg["weight_broadcast"]["bytes"] > 200e6 and cfg["weight_broadcast"]["ms"] > 0 and cfg["weight_broadcast"]["backend"] == "gloo"
    assert cfg["per_rank_images_per_s"]["min"] > 0


def test_bench_dist_selftest_runs_the_rccl_start_up_path_on_one_gpu():
    """the `nccl` (= RCCL) branch of the N > 1 path had never executed anywhere (the one-GPU functional test is
    gloo: RCCL refuses two ranks on one device).  --dist-selftest runs it with ONE rank: communicator set-up, the broadcast of
    the packed weights on the device, the receiver's import on a second handle (bytes compared), one all-reduce."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--dist-selftest", "--no-cpu-baseline", "--no-also", "--parity-dtype", "none", "--profile-steps", "1"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    wb = line["config"]["weight_broadcast"]
    assert line["n_gpus"] == 1 and wb["backend"] == "nccl" and wb["bytes"] > 200e6 and "selftest" in wb
