"""Multi-GPU start-up for data-parallel inference.

The path shards by image: every rank owns a full replica and independent batches, and the
steady-state loop has NO collective.  The only exchange is at start-up: rank 0 folds / packs
the weights once and the packed blob (~244 MB) is broadcast with torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of a global batch for `rank`; remainder goes to the low ranks."""
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _barrier(device_index: int):
    """dist.barrier() on THIS rank's device.  With the nccl (= RCCL) backend a bare barrier() makes torch guess the device as
    rank % n_gpus unless init_process_group got device_id; a caller whose device_index differs from that guess would create a
    communicator on the wrong GPU, which torch warns can hang."""
    if dist.get_backend() == "nccl":
        dist.barrier(device_ids=[int(device_index)])
    else:
        dist.barrier()


def broadcast_blob(blob: Optional[torch.Tensor], nbytes: int, device: torch.device, src: int = 0) -> torch.Tensor:
    """Broadcasts a uint8 blob of known size from `src`; other ranks pass blob=None."""
    if dist.get_rank() == src:
        assert blob is not None and blob.numel() == nbytes and blob.dtype == torch.uint8
        t = blob.to(device).contiguous()
    else:
        t = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(t, src=src)
    return t


def broadcast_fp8_calibration(eng, x: Optional[torch.Tensor], device: torch.device, src: int = 0):
    """fp8 dtype: `src` measures the activation scales of the e4m3 tensors on its batch `x` (Engine.calibrate_fp8) and every
    rank installs the same 128 power-of-two scales -- one small broadcast at start-up, none in the loop."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        eng.calibrate_fp8(x)
        return
    t = torch.ones(128, dtype=torch.float32, device=device)
    if dist.get_rank() == src:
        eng.calibrate_fp8(x)
        s, _ = eng.fp8_calibration()
        t[: len(s)] = torch.from_numpy(s).to(device)
    dist.broadcast(t, src=src)
    if dist.get_rank() != src:
        eng.set_fp8_calibration(t.cpu().numpy())


def build_replicated_engine(state_dict_fn, num_channels: int, max_batch: int, dtype: str, device_index: int, src: int = 0,
                            dual: bool = False, x3_groups=0, backbone: str = "vitb_rn50_384", calib_x=None,
                            selftest: bool = False):
    """Every rank gets an Engine with identical packed weights; only `src` runs the host-side
    fold/pack (state_dict_fn() is called on `src` only).

    dtype 'fp8': the activation scales of the e4m3 tensors are data, and replicas must agree on them -- `calib_x` (a
    representative batch on `src`'s device; other ranks may pass None) is measured on `src` and the scales travel with
    the weights (one more 512-byte broadcast).  Without `calib_x` the engine stays uncalibrated: calibrate one rank and
    call broadcast_fp8_calibration, or every rank would calibrate on its own shard and the replicas would diverge
    (the model warns when it has to calibrate implicitly).

    selftest (bench.py --dist-selftest): with ONE rank and an initialised process group the exchange still runs -- export,
    broadcast through the backend (RCCL when "nccl": communicator set-up and one collective on the device blob), and the
    RECEIVER's half on a second handle (import_packed) whose re-exported blob must equal the sender's byte for byte.  That is
    the N > 1 start-up path executed end to end on a one-GPU box; `eng.replication["selftest"]` records it."""
    from .engine import Engine
    eng = Engine(num_channels=num_channels, max_batch=max_batch, dtype=dtype, device_id=device_index, dual=dual,
                 x3_groups=x3_groups, backbone=backbone)
    device = torch.device("cuda", device_index)
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not selftest):
        eng.load_state_dict(state_dict_fn())
        if dtype == "fp8" and calib_x is not None:
            eng.calibrate_fp8(calib_x)
        return eng
    import time
    if dist.get_world_size() == 1:   # selftest: sender and receiver are this rank
        eng.load_state_dict(state_dict_fn())
        blob = eng.export_packed()
        torch.cuda.current_stream().synchronize()
        _barrier(device_index)
        t0 = time.perf_counter()
        got = broadcast_blob(blob, eng.packed_bytes, device, src)
        torch.cuda.synchronize(device)
        ms = 1e3 * (time.perf_counter() - t0)
        rx = Engine(num_channels=num_channels, max_batch=1, dtype=dtype, device_id=device_index, dual=dual, x3_groups=x3_groups,
                    backbone=backbone)
        rx.import_packed(got)
        same = bool(torch.equal(rx.export_packed(), blob))
        rx.close()
        if not same:
            raise RuntimeError("dist selftest: the blob re-exported by the receiving handle differs from the sender's")
        eng.replication = {"bytes": int(eng.packed_bytes), "ms": round(ms, 3), "backend": dist.get_backend(), "world": 1, "src": src,
                           "selftest": "one rank: export -> broadcast -> import on a second handle -> re-export == sent blob"}
        if dtype == "fp8" and calib_x is not None:
            eng.calibrate_fp8(calib_x)
        return eng
    if dist.get_rank() == src:
        eng.load_state_dict(state_dict_fn())
        blob = eng.export_packed()
        torch.cuda.current_stream().synchronize()
        _barrier(device_index)   # the receivers wait here for the host-side fold / pack, not inside the timed broadcast
        t0 = time.perf_counter()
        broadcast_blob(blob, eng.packed_bytes, device, src)
    else:
        _barrier(device_index)
        t0 = time.perf_counter()
        blob = broadcast_blob(None, eng.packed_bytes, device, src)
        eng.import_packed(blob)
    torch.cuda.synchronize(device)
    # what the start-up exchange was (bench.py prints it): bytes, wall time on this rank, backend ("nccl" == RCCL on ROCm)
    eng.replication = {"bytes": int(eng.packed_bytes), "ms": round(1e3 * (time.perf_counter() - t0), 3),
                       "backend": dist.get_backend(), "world": dist.get_world_size(), "src": src}
    if dtype == "fp8":
        # every rank takes part in the (collective) decision whether scales travel: src says whether it has a calibration batch
        has = torch.tensor([1 if (dist.get_rank() == src and calib_x is not None) else 0], dtype=torch.int32, device=device)
        dist.broadcast(has, src=src)
        if int(has.item()):
            broadcast_fp8_calibration(eng, calib_x if dist.get_rank() == src else None, device, src)
    return eng
