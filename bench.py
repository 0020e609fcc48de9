#!/usr/bin/env python3
"""bench.py -- DPT-Hybrid-384 surface-normal inference throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one forward of the hot path (dptx_forward through the C ABI) over one batch of 32
synthetic 384x384 images that are already resident in HBM (BASELINE.json configs[1]: DPT-Hybrid-384
surface normals, batch 32, bf16, one MI355X).  With N > 1 every rank owns a replica (weights
packed once on rank 0 and broadcast over RCCL/xGMI) and its own batch; there is no collective in
the timed loop (weak scaling).  Rank 0 prints ONE JSON line.

The line also carries
  roofline     : MFMA roofline of the dominant kernel family (the implicit-GEMM MFMA kernel that
                 runs every conv / linear).  achieved = algorithmic FLOPs routed through it per
                 forward / its summed launch durations, measured with HIP events on the forward's
                 stream in profiled forwards run right after the timed region.
  cpu_baseline : the CPU fp32 oracle (oracle/dpt_oracle.py, a port of the reference forward) timed
                 on this host's cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import socket
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# "dual" = BASELINE.json configs[4]: both tasks from one encoder pass, per image (185.29 GMAC)
GFLOP_PER_IMAGE = {"normal": 255.25, "depth": 255.23, "dual": 370.58}          # SURVEY.md 8d (algorithmic)
GEMM_GMAC_PER_IMAGE = {"normal": 121.487, "depth": 121.478, "dual": 179.153}  # A.6: convs + linears (attention excluded)
# --backbone vitl16_384 (DPT-Large; not the headline configuration): 258.21 GMAC/image, 16.36 of them attention
GFLOP_LARGE = {"normal": 516.43, "depth": 516.41}
GEMM_GMAC_LARGE = {"normal": 241.855, "depth": 241.845}
PEAK_TFLOPS = 2500.0                                           # dense bf16/fp16 MFMA, MI355X_MICROARCH.md


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_command(n_gpus, argv, port=None):
    """The command `python bench.py --gpus N ...` re-executes itself as when N > 1 and no launcher set WORLD_SIZE: one rank per
    GPU under torch.distributed.run (the form the driver's scaling run uses; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come
    from it).  Rendezvous on 127.0.0.1 -- the container hostname may not resolve."""
    if port is None:
        port = free_port()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def rank_cpu_set(local_rank, world, cpus):
    """The host cores of one rank: the process's allowed CPUs split into `world` contiguous shares (remainder to the low ranks).
    Eight Python hosts on one node otherwise migrate over all cores and fight for them while they enqueue ~200 launches per
    forward each; a share per rank keeps every rank's launch thread where its caches are."""
    cpus = sorted(cpus)
    if world <= 1 or len(cpus) < world:
        return cpus
    base, rem = divmod(len(cpus), world)
    lo = local_rank * base + min(local_rank, rem)
    return cpus[lo: lo + base + (1 if local_rank < rem else 0)]


FAMILY_KERNELS = ("gemm_glds_kernel", "gemm_pp_kernel", "gemm_pp2_kernel", "gemm_reg_kernel", "stem_conv_kernel", "head_tail_kernel",
                  "head_tail_x3_kernel")


def measure_traffic(args, lib_hash):
    """Two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, --kernel-trace only, as MI355X_MICROARCH.md's HBM
    section prescribes) over `bench.py --traffic-child` (whole-batch forwards, one at a time on one stream), summed per kernel
    family from the rocpd databases.  Forwards are COUNTED in the trace (launches of the first kernel of a forward), not assumed.
    Returns the dict that profiles/r0N_pmc_traffic.json holds, or None."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        print("measure_traffic: rocprofv3 not found", file=sys.stderr)
        return None
    tmp = tempfile.mkdtemp(prefix="dptx_traffic_", dir="/tmp")
    env = dict(os.environ, DPTX_STREAMS="1", TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    child = [sys.executable, os.path.abspath(__file__), "--traffic-child", "--steps", "3", "--batch", str(args.batch), "--dtype", args.dtype,
             "--task", args.task, "--backbone", args.backbone, "--io", args.io]
    sums = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            r = subprocess.run([rocprof, "--kernel-trace", "--pmc", counter, "-d", d, "-o", "r", "--"] + child, cwd="/tmp", env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                print(f"measure_traffic: rocprofv3 --pmc {counter} failed (rc {r.returncode})\n{r.stdout[-2000:]}", file=sys.stderr)
                return None
            cur = sqlite3.connect(dbs[0]).cursor()
            fam = allk = 0.0
            for name, cname, val in cur.execute("select name, counter_name, counter_value from pmc_events"):
                if cname != counter:
                    continue
                allk += val
                if any(f in name for f in FAMILY_KERNELS):
                    fam += val
            # forwards and family launches are COUNTED in the kernel trace of the same run (a counter may have several rows per dispatch)
            names = [r[0] for r in cur.execute("select name from kernels")]
            forwards = sum(1 for n in names if "stem_conv_kernel" in n or "patchify16_kernel" in n)
            gemm_launches = sum(1 for n in names if any(f in n for f in FAMILY_KERNELS))
            sums[counter] = (fam, allk, forwards, gemm_launches)
    except Exception as ex:   # the headline must not die with an optional measurement
        print(f"measure_traffic: {type(ex).__name__}: {ex}", file=sys.stderr)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    (f_fam, f_all, fw, nl), (w_fam, w_all, fw2, _) = sums["FETCH_SIZE"], sums["WRITE_SIZE"]
    if fw < 1 or fw2 < 1:
        print("measure_traffic: no forward found in the trace", file=sys.stderr)
        return None
    corr = 2.0   # gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM)
    fam_bytes = corr * f_fam * 1024.0 / fw + w_fam * 1024.0 / fw2
    all_bytes = corr * f_all * 1024.0 / fw + w_all * 1024.0 / fw2
    launches = nl / fw
    return {"source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes, DPTX_STREAMS=1, bench.py "
                      f"--traffic-child B={args.batch} {args.dtype} {args.task}",
            "library": lib_hash, "forwards": fw, "gemm_family_fetch_kb_raw_per_forward": f_fam / fw,
            "gemm_family_write_kb_per_forward": w_fam / fw2, "gfx950_fetch_correction": corr,
            "gemm_family_hbm_bytes_per_forward": fam_bytes, "all_kernels_hbm_bytes_per_forward": all_bytes,
            "gemm_launches_per_forward": launches, "gemm_family_hbm_bytes_per_launch": fam_bytes / launches}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--task", default="normal", choices=["normal", "depth", "dual"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "bf16x3", "fp16x3", "mixed", "fp8"])
    ap.add_argument("--x3-groups", default="", help="dtype mixed: layer groups that run 3 MFMAs per product, e.g. resnet+embed "
                                                    "(default: everything but the ViT blocks)")
    ap.add_argument("--parity-dtype", default="mixed", help="mode timed and checked against the fp32 oracle next to --dtype "
                                                            "(N = 1 only); 'none' skips it")
    ap.add_argument("--parity-x3-groups", default="")
    ap.add_argument("--io", default="fp32", choices=["fp32", "bf16", "fp16"], help="element type of the caller-side image and "
                    "result tensors (SURVEY.md 8d config 2 feeds bf16; fp32 is the reference's drop-in convention)")
    ap.add_argument("--backbone", default="vitb_rn50_384", choices=["vitb_rn50_384", "vitl16_384"],
                    help="vitb_rn50_384 = DPT-Hybrid (BASELINE.json's configuration, the default); vitl16_384 = DPT-Large")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the short secondary measurements (depth, dual-task bf16 / fp8) "
                                                            "that the default N = 1 normal-head run appends under 'also'")
    ap.add_argument("--inflight", type=int, default=None,
                    help="batches in flight per GPU in the HEADLINE loop.  Default (not given): 1 = what the drop-in call `model(x)` / "
                         "`torch.hub.load(...)(x)` delivers -- one forward at a time, stream-ordered.  n > 1: omnidata_amd/pipeline.py "
                         "ForwardPipeline (that many handles, one stream each, one shared copy of the weights; every step is still one "
                         "forward of one batch of --batch images; batch latency ~ n x ms_per_step, n arenas).  At N = 1 BOTH schedules "
                         "are timed in the same process and reported as value_inflight1 / value_inflight2; `value` is the headline's")
    ap.add_argument("--no-schedule-ab", action="store_true", help="skip the other schedule's loop (rocprofv3 passes: one schedule per trace)")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--measure-traffic", action="store_true",
                    help="N = 1: measure roofline.traffic in THIS run -- two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate "
                         "child processes of this script, a few single-stream forwards each; + ~2 min) instead of reading the committed "
                         "measurement; --traffic-out writes the result (stamped with the library's source hash) for later runs")
    ap.add_argument("--traffic-out", default=None, help="with --measure-traffic: JSON file to write (e.g. profiles/r06_pmc_traffic.json)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--profile-dump", default=None, help="write per-launch CSV of one profiled forward here")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL; default) or gloo (functional test of the N>1 path)")
    ap.add_argument("--share-gpu", action="store_true", help="debug: all ranks use cuda:0 (needs --dist-backend gloo)")
    ap.add_argument("--dist-selftest", action="store_true",
                    help="N = 1 only: initialise the process group anyway (backend --dist-backend, nccl = RCCL) and run the N > 1 "
                         "start-up path -- packed-weight export, broadcast, import on a second handle, byte comparison -- plus one "
                         "all-reduce, on this one GPU; the line's config.weight_broadcast records it")
    ap.add_argument("--rendezvous-check", action="store_true",
                    help="launch the N ranks, count them with one all-reduce (gloo, no GPU touched), print {n_gpus, ranks_seen} "
                         "and exit: the CPU test of the --gpus N launch path")
    args = ap.parse_args()
    inflight_explicit = args.inflight is not None
    if args.inflight is None:
        args.inflight = 1

    # --gpus N is the number of ranks.  Under a launcher (torch.distributed.run: the driver's N > 1 form) WORLD_SIZE is set and
    # must agree; a plain `python bench.py --gpus N` starts the N ranks itself by re-executing under torch.distributed.run.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.stdout.flush()
        os.execv(sys.executable, spawn_command(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE thing: rank 0's JSON line.  Libraries print there too (RCCL's version banner goes to the C stdout and
    # is flushed at exit, i.e. AFTER a Python print -- found in this round's evidence run), so file descriptor 1 is pointed at
    # stderr for the whole run and the line is written to the saved descriptor at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; the line would "
                         f"report the wrong n_gpus -- pass --gpus {world} (or launch {args.gpus} ranks)")
    # one share of the host's cores per rank (N > 1): see rank_cpu_set
    affinity = None
    if world > 1 and hasattr(os, "sched_setaffinity"):
        try:
            mine = rank_cpu_set(local_rank, world, os.sched_getaffinity(0))
            os.sched_setaffinity(0, mine)
            affinity = [min(mine), max(mine), len(mine)]
        except OSError:
            affinity = None
    if args.rendezvous_check:
        import torch.distributed as dist
        seen = 1
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo")
            t = torch.ones(1, dtype=torch.int64)
            dist.all_reduce(t)
            seen = int(t.item())
            ncpu = torch.tensor([len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 0], dtype=torch.int64)
            lo = ncpu.clone()
            dist.all_reduce(ncpu)                         # disjoint shares add up to the cores the launcher had
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.barrier()
            dist.destroy_process_group()
            if rank == 0:
                emit({"n_gpus": world, "ranks_seen": seen, "rendezvous_check": True, "cpus_pinned_total": int(ncpu.item()),
                      "cpus_per_rank_min": int(lo.item())})
            return
        if rank == 0:
            emit({"n_gpus": world, "ranks_seen": seen, "rendezvous_check": True})
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path exists only as HIP kernels (no CPU fallback)")
    if not args.share_gpu and world > torch.cuda.device_count():
        raise SystemExit(f"bench.py: {world} ranks but {torch.cuda.device_count()} visible GPUs (one rank per GPU; "
                         "--share-gpu --dist-backend gloo is the one-GPU functional test of the N > 1 path)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1 or args.dist_selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.dist_backend)

    from omnidata_amd.build import build
    if rank == 0 or not os.path.exists(os.path.join(ROOT, "omnidata_amd", "libdptx.so")):
        build()
    if world > 1:
        dist.barrier()
    from omnidata_amd.dist import build_replicated_engine
    from omnidata_amd.weights import random_dual_state_dict, random_state_dict, synthetic_input

    dual = args.task == "dual"
    large = args.backbone == "vitl16_384"
    if large and dual:
        raise SystemExit("the dual-task model is DPT-Hybrid")
    if large:
        GFLOP_PER_IMAGE.update(GFLOP_LARGE)
        GEMM_GMAC_PER_IMAGE.update(GEMM_GMAC_LARGE)
    C = 1 if args.task == "depth" else 3
    make_sd = (lambda: random_dual_state_dict(0)) if dual else (lambda: random_state_dict(0, C, backbone=args.backbone))
    eng = build_replicated_engine(make_sd, C, args.batch, args.dtype, local_rank, dual=dual, x3_groups=args.x3_groups,
                                  backbone=args.backbone, selftest=args.dist_selftest and world == 1)
    if args.dist_selftest and world == 1:   # one collective besides the broadcast
        one = torch.ones(1, dtype=torch.int64, device=device)
        dist.all_reduce(one)
        assert int(one.item()) == 1
    io_dt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[args.io]
    x = synthetic_input(1000 + rank, args.batch, "normal" if dual else args.task).to(device).to(io_dt)
    y = torch.empty(args.batch, C, 384, 384, dtype=io_dt, device=device)
    if dual:  # one step = one encoder pass + both decoders on the batch
        y2 = torch.empty(args.batch, 1, 384, 384, dtype=io_dt, device=device)
        eng.forward = lambda x_, out=None: eng.forward_dual(x_, out_normal=out, out_depth=y2)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.dtype == "fp8":  # per-tensor activation scales of the e4m3 copies: rank 0 measures them on its batch
        from omnidata_amd.dist import broadcast_fp8_calibration
        broadcast_fp8_calibration(eng, x, device)

    if args.traffic_child:   # under rocprofv3 --pmc (see measure_traffic): whole-batch forwards, one at a time, then exit
        for _ in range(max(1, args.steps)):
            eng.forward(x, out=y)
        torch.cuda.synchronize()
        return

    # the timed loop: K steps, each ONE forward of one batch; with --inflight n > 1 consecutive steps go to n handles on n
    # streams (one copy of the weights) and overlap -- every step is enqueued inside the timed region and complete at its end
    from omnidata_amd.pipeline import ForwardPipeline
    from omnidata_amd.telemetry import GpuTelemetry
    tele = GpuTelemetry(local_rank) if rank == 0 else None
    tele_idle = tele.read_once() if tele is not None else None
    tele_log = {}

    def timed(e_, inflight, steps, warmup, x_, y_, dual_=False, y2_=None, tag=None):
        """seconds for `steps` forwards of x_ on engine e_ (weights loaded) with `inflight` forwards in flight, all finite?
        tag: record the GPU's clock / power / temperature over the timed region under tele_log[tag] (sampling thread, no GPU work)"""
        pipe = ForwardPipeline.from_engine(e_, depth=inflight) if inflight > 1 else None
        ys_ = [y_] + [torch.empty_like(y_) for _ in range(inflight - 1)]
        y2s_ = ([y2_] + [torch.empty_like(y2_) for _ in range(inflight - 1)]) if dual_ else None

        def one(i):
            k = i % inflight
            if pipe is None:
                if dual_:
                    e_.forward_dual(x_, out_normal=y_, out_depth=y2_)
                else:
                    type(e_).forward(e_, x_, out=y_)
            elif dual_:
                pipe.submit(x_, out=ys_[k], out_depth=y2s_[k])
            else:
                pipe.submit(x_, out=ys_[k])
        for i in range(max(warmup, inflight)):
            one(i)
        sync_all()
        if tag is not None and tele is not None:
            tele.start()
        t_ = time.perf_counter()
        for i in range(steps):
            one(i)
        sync_all()
        dt_ = time.perf_counter() - t_
        if tag is not None and tele is not None:
            tele_log[tag] = tele.stop().summary()
        ok_ = all(bool(torch.isfinite(t.float()).all()) for t in ys_)
        if pipe is not None:
            pipe.close()
        return dt_, ok_

    # n forwards in flight need n activation arenas next to the model's own (ForwardPipeline.from_engine): refuse what does not
    # fit instead of dying in hipMalloc halfway through the run
    def pipeline_fits(n):
        free, _total = torch.cuda.mem_get_info(device)
        arena = eng.workspace_bytes - eng.packed_bytes
        return n * arena + (256 << 20) <= free, n * arena, free
    if args.inflight > 1:
        ok_fit, need_b, free_b = pipeline_fits(args.inflight)
        if not ok_fit:
            raise SystemExit(f"bench.py: --inflight {args.inflight} needs {need_b / 2**30:.1f} GiB of activation arenas on top of what is "
                             f"allocated, {free_b / 2**30:.1f} GiB of HBM are free on {device}: lower --inflight or --batch")
    # The other schedule first (rank 0, N = 1), the headline last: both schedules on THIS box in THIS process.  (First, so
    # that the engine's internal streams exist before the pipeline's: measured the other way round -- lease r5l3 -- the
    # one-forward-at-a-time loop ran at 1993 instead of ~2600 images/s; HIP maps streams onto a small number of hardware queues.)
    schedule_ab = None
    if rank == 0 and world == 1 and not args.no_schedule_ab and (args.inflight > 1 or pipeline_fits(2)[0]):
        other = 1 if args.inflight > 1 else 2
        dt_o, _ = timed(eng, other, args.steps, args.warmup, x, y, dual, y2 if dual else None, tag=f"inflight{other}")
        schedule_ab = {"inflight": other, "value": round(args.batch * args.steps / dt_o, 2), "unit": "images/s",
                       "ms_per_step": round(1e3 * dt_o / args.steps, 3),
                       "note": "same process, same weights, measured right before the headline loop"}
    elapsed, finite = timed(eng, args.inflight, args.steps, args.warmup, x, y, dual, y2 if dual else None, tag=f"inflight{args.inflight}")
    assert finite
    # what proves that N ranks ran, each on its own GPU: an all-reduce of 1, every rank's device index and own clock
    ranks_seen, rank_devices, per_rank = 1, [local_rank], [args.batch * args.steps / elapsed]
    if world > 1:
        own = elapsed
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        one = torch.ones(1, dtype=torch.int64, device=device)
        dist.all_reduce(one)
        ranks_seen = int(one.item())
        g = torch.zeros(world, 2, dtype=torch.float64, device=device)   # a gather as a sum of one-hot rows (gloo has no CUDA all_gather)
        g[rank, 0], g[rank, 1] = float(torch.cuda.current_device()), args.batch * args.steps / own
        dist.all_reduce(g)
        rank_devices = [int(v) for v in g[:, 0].tolist()]
        per_rank = [float(v) for v in g[:, 1].tolist()]
    # ---- roofline of the dominant kernel family: HIP events around every launch, separate passes
    roofline = None
    breakdown = None
    if rank == 0:
        eng.set_profiling(True)
        acc = {}
        for _ in range(max(1, args.profile_steps)):
            eng.forward(x, out=y)
            torch.cuda.synchronize()
            for k, (ms, n, macs) in eng.profile().items():
                a = acc.setdefault(k, [0.0, n, macs])
                a[0] += ms
        if args.profile_dump:
            eng.profile_dump(args.profile_dump)
        eng.set_profiling(False)
        P = max(1, args.profile_steps)
        breakdown = {k: {"ms_per_step": round(v[0] / P, 4), "launches": v[1], "executed_gmac_per_image": round(v[2] / 1e9, 3)}
                     for k, v in acc.items()}
        gemm_ms = acc["gemm"][0] / P
        gemm_flop = 2.0 * GEMM_GMAC_PER_IMAGE[args.task] * 1e9 * args.batch
        achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12
        # HBM traffic of the family per launch.  PMC counters cannot be read inside this process: either --measure-traffic
        # runs the two rocprofv3 passes as child processes NOW, or the newest committed measurement (profiles/r0N_pmc_traffic.json)
        # is used -- but only if it was taken on THIS library (its "library" stamp equals dptx_version()'s src hash); a figure
        # measured on other kernels is reported as null with the reason, not passed off as this run's
        lib_hash = eng.lib.dptx_version().decode().rsplit("src=", 1)[-1]
        traffic, traffic_src = None, None
        if args.measure_traffic and world == 1:
            tj = measure_traffic(args, lib_hash)
            if tj is not None:
                traffic = round(tj["gemm_family_hbm_bytes_per_launch"])
                traffic_src = "measured in this run: " + tj["source"]
                if args.traffic_out:
                    with open(args.traffic_out, "w") as f:
                        json.dump(tj, f, indent=1)
            else:
                traffic_src = "--measure-traffic failed (rocprofv3 unavailable or its output unreadable): see stderr"
        else:
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
            if cands and args.dtype == "bf16" and not large and args.task == "normal":
                tj = json.load(open(cands[-1]))   # newest committed round
                if tj.get("library") == lib_hash:
                    traffic = round(tj["gemm_family_hbm_bytes_per_launch"] * args.batch / 32.0)
                    traffic_src = f"committed measurement {os.path.basename(cands[-1])} on this library ({lib_hash}), B=32"
                else:
                    traffic_src = (f"null: {os.path.basename(cands[-1])} was measured on library {tj.get('library', '(unstamped, rounds 1-5)')}, "
                                   f"this run's is {lib_hash}; run `python bench.py --measure-traffic` for a figure of this build")
        # the fraction is quoted for the single-pass 16-bit modes only: fp8 runs part of the family on the 2x-rate e4m3 MFMA
        # and the 3-MFMA modes execute up to three products per algorithmic one, so "achieved / bf16 peak" would not be a
        # utilisation figure there
        single_pass = args.dtype in ("bf16", "fp16")
        roofline = {"bound": "mfma", "kernel": "dptx::gemm_kernel (implicit-GEMM MFMA, all conv/linear launches)",
                    "achieved": round(achieved, 2), "peak": PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_TFLOPS, 4) if single_pass else None,
                    "frac_note": None if single_pass else "not quoted: algorithmic FLOP/s of a mode that runs part of the family on "
                                 "the fp8 MFMA (2x peak) or with 3 MFMAs per product is not a utilisation of the bf16 peak",
                    "traffic": traffic, "traffic_source": traffic_src,
                    "traffic_note": "HBM bytes per gemm launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 / launches from rocprofv3 PMC passes "
                                    "(separate passes; gfx950 FETCH_SIZE x2: MI355X_MICROARCH.md HBM); algorithmic min ~ A+C+W bytes "
                                    "= 16.64 GB / 130 launches = 128 MB per launch at B=32",
                    "launches_per_step": acc["gemm"][1], "avg_launch_ms": round(gemm_ms / max(1, acc["gemm"][1]), 5),
                    "algorithmic_gflop_per_step": round(gemm_flop / 1e9, 1),
                    "executed_gflop_per_step": round(2 * acc["gemm"][2] * args.batch / 1e9, 1)}

    # ---- CPU baseline: the fp32 oracle on this host, bounded sample
    cpu_baseline = None
    parity = None
    oracle_ref = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.dpt_oracle import dpt_forward, dpt_forward_dual, dpt_forward_vitl16
        sd = make_sd()
        if large:
            dpt_forward = dpt_forward_vitl16  # noqa: F811
        if dual:
            dpt_forward = dpt_forward_dual  # noqa: F811  (the reference forward twice, encoder weights tied)
        xc = synthetic_input(1000, 4, "normal" if dual else args.task)
        ncpu = os.cpu_count() or 1
        # oneDNN/MKL at batch 4 stop scaling (and then collapse) well before 256 threads: probe a few
        # thread counts on one batch each and keep the fastest for the timed sample
        best, best_t = None, None
        tried = {}
        for nt in sorted({min(ncpu, n) for n in (16, 32, 64, 128, 256)}):
            torch.set_num_threads(nt)
            dpt_forward(sd, xc[:1])
            t1 = time.perf_counter()
            dpt_forward(sd, xc)
            dt1 = time.perf_counter() - t1
            tried[nt] = round(xc.shape[0] / dt1, 3)
            if best_t is None or dt1 < best_t:
                best, best_t = nt, dt1
            if dt1 > 20:
                break
        torch.set_num_threads(best)
        t1 = time.perf_counter()
        n_img = 0
        while True:
            oracle_ref = dpt_forward(sd, xc)
            n_img += xc.shape[0]
            if time.perf_counter() - t1 > 10.0 or n_img >= 64:
                break
        dt_cpu = time.perf_counter() - t1
        cpu_baseline = {"value": round(n_img / dt_cpu, 3), "unit": "images/s", "cores": best, "host_logical_cpus": ncpu,
                        "kind": "port",
                        "threads_tried_images_per_s": tried,
                        "threads_note": "north_star asks for all host cores: every count in threads_tried was timed on one batch of 4 "
                                        "and the fastest kept -- oneDNN at this batch size is slower on more threads, so `cores` is the "
                                        "count that gives the CPU its best number, of host_logical_cpus available",
                        "kind_note": "oracle/dpt_oracle.py: functional restatement of the reference forward, pinned at 0.0 against "
                                     "the reference's unmodified modules + timm shim by oracle/validate_vs_reference.py",
                        "sample": f"{n_img} images (batches of 4) of the same synthetic 384x384 workload, fp32, "
                                                  f"best of {sorted(tried)} threads"}

    # ---- parity: the benched dtype AND the parity mode against the fp32 oracle on the cpu_baseline sample, and the parity
    # mode's own throughput (same batch, same timing protocol), so that "matches the reference" and "img/s" are stated
    # for the same configuration in one line
    if rank == 0 and world == 1 and oracle_ref is not None:
        def max_abs(e_, outs_ref):
            outs = e_.forward_dual(xc.to(device)) if dual else (e_.forward(xc.to(device)),)
            refs = outs_ref if dual else (outs_ref,)
            return max(float((o.squeeze(1).cpu() - r.reshape(o.squeeze(1).shape)).abs().max()) for o, r in zip(outs, refs))
        eng.forward = type(eng).forward.__get__(eng)  # undo the dual lambda for the 4-image check
        parity = {"sample": f"{xc.shape[0]} images vs the fp32 CPU oracle", "tolerance_north_star": 1e-3,
                  "benched_dtype": {"dtype": args.dtype, "max_abs": round(max_abs(eng, oracle_ref), 6)}}
        if args.parity_dtype != "none" and args.parity_dtype != args.dtype:
            from omnidata_amd.engine import Engine
            pe = Engine(num_channels=C, max_batch=args.batch, dtype=args.parity_dtype, device_id=local_rank, dual=dual,
                        x3_groups=args.parity_x3_groups, backbone=args.backbone)
            pe.load_state_dict(sd)
            pm = max_abs(pe, oracle_ref)
            pdt1, _ = timed(pe, 1 if args.inflight > 1 else 2, args.steps, args.warmup, x, y, dual, y2 if dual else None)
            pdt, _ = timed(pe, args.inflight, args.steps, args.warmup, x, y, dual, y2 if dual else None)   # the headline's schedule
            parity["parity_mode"] = {"dtype": args.parity_dtype,
                                     "x3_groups": args.parity_x3_groups or "default (all layer groups but the ViT blocks)",
                                     "value": round(args.batch * args.steps / pdt, 2), "unit": "images/s",
                                     "ms_per_step": round(1e3 * pdt / args.steps, 3), "max_abs": round(pm, 6),
                                     "meets_1e-3": bool(pm < 1e-3), "inflight": args.inflight,
                                     "schedule_ab": {"inflight": 1 if args.inflight > 1 else 2,
                                                     "value": round(args.batch * args.steps / pdt1, 2)},
                                     "value_inflight1": round(args.batch * args.steps / (pdt if args.inflight == 1 else pdt1), 2),
                                     "value_inflight2": round(args.batch * args.steps / (pdt1 if args.inflight == 1 else pdt), 2) if args.inflight <= 2 else None}
            if args.parity_dtype == "mixed" and not args.parity_x3_groups:
                # the measured option outside the default table (include/dptx.h dptx_set_layer_precision): the first head
                # convolution on two MFMAs -- weights exact, input rounded once
                for k in (("scratch.output_conv.0.weight", "depth.scratch.output_conv.0.weight") if dual else ("scratch.output_conv.0.weight",)):
                    pe.set_layer_precision(k, 2)
                pm2 = max_abs(pe, oracle_ref)
                pdt, _ = timed(pe, args.inflight, args.steps, args.warmup, x, y, dual, y2 if dual else None)
                parity["parity_mode_head0_2mfma"] = {"dtype": "mixed", "layer_precision": {"scratch.output_conv.0.weight": 2},
                                                     "value": round(args.batch * args.steps / pdt, 2), "unit": "images/s",
                                                     "ms_per_step": round(1e3 * pdt / args.steps, 3), "max_abs": round(pm2, 6),
                                                     "meets_1e-3": bool(pm2 < 1e-3),
                                                     "note": "not the default: worst of a 32-image batch 7.6e-4 (default table 6.3e-4), tests/test_gpu_mixed.py"}
            pe.close()

    # ---- BASELINE.json configs[2] and configs[4] next to the headline (configs[1]): short runs of the depth head and of the
    # dual-task forward (bf16 and with the fp8 decoder) under the same protocol, so that the driver's one default call sees them
    also = None
    if rank == 0 and world == 1 and not args.no_also and args.task == "normal" and not large and args.dtype == "bf16":
        from omnidata_amd.engine import Engine
        also = []
        # (dtype2 "fp8": the default preset -- six decoder convolutions on e4m3, within 2 x the bf16 mode's angular error;
        #  "fp8_all": all 19 eligible ones, round 3's lossy mode; "mixed": the parity mode of the dual-task model)
        # "fp8_vit" (round 6): the default decoder preset + qkv / fc1 / fc2 of the ViT blocks on e4m3 (DPTX_FLAG_FP8_VIT: 59 of the
        # dual forward's 185 GMAC); "fp8_all_vit": all 19 decoder convolutions + the ViT linears
        FP8_FLAGS = {"fp8": 0, "fp8_all": 16, "fp8_vit": 32, "fp8_all_vit": 48}
        for task2, dtype2, cfg_i in (("depth", "bf16", 2), ("dual", "bf16", 4), ("dual", "fp8", 4), ("dual", "fp8_vit", 4), ("dual", "fp8_all", 4),
                                     ("dual", "fp8_all_vit", 4), ("dual", "mixed", 4)):
            d2 = task2 == "dual"
            C2 = 1 if task2 == "depth" else 3
            e2 = Engine(num_channels=C2, max_batch=args.batch, dtype="fp8" if dtype2.startswith("fp8") else dtype2, device_id=local_rank, dual=d2,
                        flags=FP8_FLAGS.get(dtype2, 0))
            e2.load_state_dict(random_dual_state_dict(0) if d2 else random_state_dict(0, C2))
            x2 = synthetic_input(1000, args.batch, "normal" if d2 else task2).to(device).to(io_dt)
            ya = torch.empty(args.batch, C2, 384, 384, dtype=io_dt, device=device)
            yb = torch.empty(args.batch, 1, 384, 384, dtype=io_dt, device=device)
            if dtype2.startswith("fp8"):
                e2.calibrate_fp8(x2)
            n2 = max(4, min(args.steps, 20))
            dt2o, _ = timed(e2, 1 if args.inflight > 1 else 2, n2, 3, x2, ya, d2, yb if d2 else None)   # the other schedule
            dt2, _ = timed(e2, args.inflight, n2, 3, x2, ya, d2, yb if d2 else None)    # the headline's schedule
            max_abs2 = None
            if dtype2 == "mixed" and not args.no_cpu_baseline:   # the dual-task parity mode against the oracle (2 images, both heads)
                from oracle.dpt_oracle import dpt_forward_dual, oracle_threads
                oracle_threads()
                e2.forward_dual(x2[:2], out_normal=ya[:2], out_depth=yb[:2])
                rn, rd = dpt_forward_dual(random_dual_state_dict(0), x2[:2].float().cpu())
                max_abs2 = round(max(float((ya[:2].float().cpu() - rn).abs().max()), float((yb[:2, 0].float().cpu() - rd).abs().max())), 6)
            also.append({"workload": f"DPT-Hybrid-384 {task2}, batch {args.batch}, {dtype2}, 1xMI355X (BASELINE.json configs[{cfg_i}])",
                         "task": task2, "dtype": dtype2, "value": round(args.batch * n2 / dt2, 2), "unit": "images/s", "inflight": args.inflight,
                         "value_inflight1": round(args.batch * n2 / (dt2 if args.inflight == 1 else dt2o), 2),
                         "value_inflight2": round(args.batch * n2 / (dt2o if args.inflight == 1 else dt2), 2) if args.inflight <= 2 else None,
                         "steps": n2, "ms_per_step": round(1e3 * dt2 / n2, 3),
                         "e2e_tflops_algorithmic": round(args.batch * n2 / dt2 * GFLOP_PER_IMAGE[task2] / 1e3, 1),
                         "max_abs_vs_oracle": max_abs2,
                         "accuracy_note": {"fp8": "six decoder convolutions on e4m3: mean angular error within 2 x the bf16 mode's on both "
                                                  "synthetic weight families (tests/test_gpu_fp8.py); NOT validated on the published checkpoints",
                                           "fp8_vit": "six decoder convolutions + qkv / fc1 / fc2 of the 12 ViT blocks on e4m3 (per-channel weight "
                                                      "scales, one calibrated scale per activation tensor): within 2 x the bf16 mode's mean "
                                                      "angular error on both synthetic weight families (tests/test_gpu_fp8.py, oracle/fp8_vit.py)",
                                           "fp8_all": "all 19 eligible decoder convolutions on e4m3: a lossy throughput mode (7 - 9 deg mean "
                                                      "angular error, tests/test_gpu_fp8.py)",
                                           "fp8_all_vit": "fp8_all + the ViT linears: the fastest and lossiest preset"}.get(dtype2)})
            e2.close()
            del e2, x2, ya, yb
            torch.cuda.empty_cache()
        # fp8 is a throughput mode only where it is faster than the bf16 engine it approximates: say which it is on THIS box
        dual_bf16 = next((a["value"] for a in also if a["task"] == "dual" and a["dtype"] == "bf16"), None)
        for a in also:
            if a["dtype"].startswith("fp8") and dual_bf16:
                a["vs_dual_bf16"] = round(a["value"] / dual_bf16, 3)
                gain = a["value"] / dual_bf16 - 1.0
                a["verdict"] = (f"pays on this box: {100 * gain:+.1f} % over dual bf16 (bar: >= +10 %)" if gain >= 0.10 else
                                f"does NOT pay on this box: {100 * gain:+.1f} % against dual bf16 (bar: >= +10 %) for "
                                + ("7-9 deg of" if a["dtype"].startswith("fp8_all") else "up to 2x the bf16 engine's") + " mean angular error")

    if rank == 0:
        total_images = args.batch * world * args.steps
        value = total_images / elapsed
        e2e_tflops = value * GFLOP_PER_IMAGE[args.task] / 1e3
        arch = "DPT-Large" if large else "DPT-Hybrid"
        line = {
            "metric": {"normal": f"images/sec (384x384) {arch} surface-normal inference",
                       "depth": f"images/sec (384x384) {arch} depth inference",
                       "dual": "images/sec (384x384) DPT-Hybrid dual-task normal+depth, shared encoder"}[args.task],
            "value": round(value, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            # both schedules as first-class numbers of this process (N = 1): inflight1 = what `model(x)` / the hub call delivers,
            # inflight2 = ForwardPipeline / model.forward_pipelined (two arenas, ~2x batch latency).  `value` is value_inflight<headline>
            "value_inflight1": (round(value, 2) if args.inflight == 1 else (schedule_ab or {}).get("value")) if world == 1 else None,
            "value_inflight2": (round(value, 2) if args.inflight == 2 else ((schedule_ab or {}).get("value") if args.inflight == 1 else None)) if world == 1 else None,
            "headline_schedule": (f"--inflight {args.inflight} given explicitly" if inflight_explicit else
                                  "default: inflight 1 = the drop-in call model(x), one forward at a time"),
            "telemetry": {"idle_before": tele_idle, **tele_log,
                          "note": "GPU shader clock / board power / temperature sampled from sysfs by a host thread during each timed "
                                  "loop (omnidata_amd/telemetry.py): the chip is power-managed, a number is comparable across boxes "
                                  "and rounds only together with the clock it ran at"},
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "io_dtype": args.io,
            "data": "synthetic 384x384 inputs resident in HBM; seeded random weights",
            "config": {"workload": f"{arch}-384 {args.task}, batch {args.batch}/GPU, {args.dtype}, {world}xMI355X "
                                   + ("(backbone vitl16_384: SURVEY.md 8f row 3, not a BASELINE.json configuration)" if large else
                                      f"(BASELINE.json configs[{ {'normal': 1, 'depth': 2, 'dual': 4}[args.task] }])"), "batch_per_gpu": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"replicas x{world} (no collective in the loop)",
                       "ranks_seen": ranks_seen, "rank_devices": rank_devices, "dist_backend": args.dist_backend if (world > 1 or args.dist_selftest) else None,
                       "per_rank_images_per_s": {"min": round(min(per_rank), 2), "max": round(max(per_rank), 2)},
                       "rank0_cpu_affinity": affinity,   # [first, last, count] of this rank's share of the host cores (N > 1)
                       "weight_broadcast": getattr(eng, "replication", None),
                       "schedule": (f"{args.inflight} forwards of {args.batch} images in flight per GPU: {args.inflight} handles x 1 stream, one "
                                    f"shared copy of the weights (omnidata_amd/pipeline.py); batch latency ~ {args.inflight} x ms_per_step"
                                    if args.inflight > 1 else
                                    "one forward at a time on the caller's stream (the engine's own intra-forward schedule: DESIGN.md 3)"),
                       "inflight": args.inflight, "schedule_ab": schedule_ab,
                       # inflight 1: which intra-forward schedule the handle MEASURED to be faster at its first forward
                       # (include/dptx.h dptx_tune_schedule): split = two half-batches on two internal streams
                       "engine_schedule": eng.schedule_info(),
                       # A/B switches of the library that were set in this process's environment: none
                       # in a default run -- a stray one would otherwise change the kernels under the number unseen
                       "env_switches": {k: v for k, v in sorted(os.environ.items()) if k.startswith("DPTX_")}},
            "e2e_mfma_frac": round(e2e_tflops / (PEAK_TFLOPS * world), 4) if args.dtype in ("bf16", "fp16") else None,
            "e2e_tflops_algorithmic": round(e2e_tflops, 1),
            "roofline": roofline, "cpu_baseline": cpu_baseline, "parity": parity, "also": also,
            "kernel_breakdown": None if breakdown is None else dict(breakdown, overlap_note=(
                "per-launch HIP events of profiled forwards run one at a time on ONE stream after the timed region (every launch "
                "serialised): the categories sum to the single-stream forward, which is LONGER than ms_per_step whenever the timed "
                "schedule overlaps launches (image groups, two forwards in flight)")),
        }
    if world > 1 or args.dist_selftest:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        emit(line)


if __name__ == "__main__":
    main()
