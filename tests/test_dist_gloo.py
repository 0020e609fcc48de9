"""N>1 path on CPU: world_size-2 gloo.  Rank 0 folds/packs the weights (host-only handle), the
packed blob is broadcast, rank 1 checks it equals its own packing; batch sharding covers the
global batch exactly once.  No GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from omnidata_amd.dist import _barrier, broadcast_blob, shard_range
        _barrier(0)   # the start-up barrier of build_replicated_engine (device-pinned under nccl, plain under gloo)
        from omnidata_amd.engine import Engine
        from omnidata_amd.weights import random_state_dict
        eng = Engine(num_channels=3, max_batch=2, dtype="bf16", device_id=None)   # host-only handle
        nbytes = eng.packed_bytes
        blob = None
        if rank == 0:
            eng.load_state_dict(random_state_dict(11, 3))
            blob = torch.from_numpy(eng.export_packed_host())
        got = broadcast_blob(blob, nbytes, torch.device("cpu"), src=0)
        if rank != 0:   # the receiving rank packs the same seed itself and must get identical bytes
            eng.load_state_dict(random_state_dict(11, 3))
            assert np.array_equal(got.numpy(), eng.export_packed_host())
        # sharding: every image of a global batch is owned by exactly one rank
        for gb in (64, 65, 7, 1):
            lo, hi = shard_range(gb, rank, world)
            t = torch.zeros(gb, dtype=torch.int32)
            t[lo:hi] = 1
            dist.all_reduce(t)
            assert bool((t == 1).all())
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        assert t.item() == float(world)
        q.put((rank, "ok", int(got.numel())))
    except Exception as e:  # pragma: no cover
        q.put((rank, repr(e), 0))
    finally:
        dist.destroy_process_group()


def test_packed_weight_broadcast_and_sharding_world2(built_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] == "ok" for r in res), res
    assert res[0][2] == res[1][2] > 200e6


def test_shard_range_properties():
    from omnidata_amd.dist import shard_range
    for world in (1, 2, 3, 8):
        for gb in (0, 1, 7, 32, 255, 256):
            spans = [shard_range(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def _bench(args, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    env["PYTHONPATH"] = ROOT
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                          cwd=ROOT, timeout=300)


def test_bench_gpus_n_starts_n_ranks_in_the_plain_python_form():
    """`python bench.py --gpus 2` (no launcher, WORLD_SIZE unset) must itself start two ranks.  --rendezvous-check
    stops after the ranks have counted themselves with one all-reduce, before anything touches a GPU."""
    import json
    r = _bench(["--gpus", "2", "--rendezvous-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert {k: line[k] for k in ("n_gpus", "ranks_seen", "rendezvous_check")} == {"n_gpus": 2, "ranks_seen": 2, "rendezvous_check": True}
    r = _bench(["--gpus", "1", "--rendezvous-check"])
    assert r.returncode == 0 and json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_bench_gpus_8_rendezvous_and_disjoint_cpu_shares():
    """the driver's 8-GPU form without hardware -- `python bench.py --gpus 8 --rendezvous-check` starts eight
    ranks that count themselves over gloo, and every rank has pinned itself to its own share of the host cores (disjoint shares
    add up to the cores the launcher was allowed; on this 8-CPU container that is one core per rank)."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    ncpu = len(os.sched_getaffinity(0))
    r = _bench(["--gpus", "8", "--rendezvous-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["ranks_seen"] == 8
    if ncpu >= 8:
        assert line["cpus_pinned_total"] == ncpu and line["cpus_per_rank_min"] == ncpu // 8
    # the partition itself: contiguous, disjoint, complete, remainder to the low ranks
    shares = [bench.rank_cpu_set(r_, 8, range(4, 4 + 19)) for r_ in range(8)]
    assert sum(shares, []) == list(range(4, 23)) and [len(s_) for s_ in shares] == [3, 3, 3, 2, 2, 2, 2, 2]
    assert bench.rank_cpu_set(0, 1, range(6)) == list(range(6)) and bench.rank_cpu_set(3, 8, range(4)) == list(range(4))


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    import subprocess
    env = dict(os.environ, PYTHONPATH=ROOT, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--rendezvous-check"], capture_output=True,
                       text=True, env=env, cwd=ROOT, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_spawn_command_is_the_drivers_launch_form():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.spawn_command(8, ["--gpus", "8", "--steps", "5"], port=29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "5"] and cmd[-5].endswith("bench.py")
