"""Several forwards in flight on one GPU (round 5).

A dptx handle is not re-entrant, and its default schedule runs the two halves of ONE batch in lockstep on two internal
streams, joined before the next forward may start: the latency-bound part of the network (the 99 dependent 20-80 us launches
of the ResNetV2 stages, 27 % of a forward at 8 % of the MFMA peak) then only ever meets ITSELF -- the other half's ResNetV2
stages -- never the MFMA-bound ViT / decoder launches.  `ForwardPipeline` keeps `depth` (default 2) whole-batch forwards in
flight instead: `depth` handles, each with its own activation arena and ONE internal stream, on `depth` HIP streams, reading
one shared copy of the packed weights (include/dptx.h dptx_share_packed).  Consecutive submissions go to the handles round
robin and free-run, so the phases of neighbouring forwards drift apart and the ResNetV2 stages of one run under the GEMMs of
the other, at full-batch tile counts.  Same kernels, same arithmetic: every result is bit-identical to the single-handle
forward (tests/test_gpu_pipeline.py).  Measured on one MI355X, B = 32, bf16: 2840 images/s against 2620 for the
two-halves schedule and 2513 for one stream (profiles/r05_experiments.md); a third forward in flight adds 0.5 %.

The price is latency (a batch takes about 1.8x as long from submission to result) and `depth` arenas (5.2 GB each at B = 32).

    pipe = ForwardPipeline(num_channels=3, max_batch=32, dtype="bf16", device_id=0)
    pipe.load_state_dict(sd)
    for y in pipe.map(batches):          # results in submission order, two batches in flight
        ...
    t = pipe.submit(x, out=y); ...; t.wait()     # explicit form: wait() orders the CURRENT stream behind the result

There is no CPU fallback (the engines raise without a GPU / libdptx.so).
"""
from __future__ import annotations

from typing import Iterable, Iterator, List, Optional

import torch

from .engine import IO_DTYPES, Engine, probe_stream_overlap


class Ticket:
    """One submitted forward: `out` is valid on a stream after that stream has passed `wait()` (or after `synchronize()`)."""

    def __init__(self, out, event: "torch.cuda.Event"):
        self.out = out
        self._event = event

    def wait(self, stream: Optional["torch.cuda.Stream"] = None):
        """Makes `stream` (default: the current one) wait for the result; does not block the host.  Returns the output."""
        (stream or torch.cuda.current_stream()).wait_event(self._event)
        return self.out

    def synchronize(self):
        """Blocks the host until the result is complete."""
        self._event.synchronize()
        return self.out

    def done(self) -> bool:
        return self._event.query()


class ForwardPipeline:
    """`depth` Engine handles on `depth` streams behind one submit / wait interface; see the module docstring.

    Constructor arguments are Engine's (num_channels, max_batch, dtype, device_id, dual, x3_groups, backbone, flags, max_hw ...);
    `inner_streams` is the internal stream count of each handle (1: the measured optimum with two forwards in flight);
    `check_overlap`: probe at construction that the slot streams really run concurrently (`overlap_ratio`; see ensure_overlap)."""

    def __init__(self, depth: int = 2, inner_streams: int = 1, check_overlap: bool = True, overlap_retries: int = 3, **engine_kwargs):
        if depth < 1:
            raise ValueError("depth >= 1")
        if engine_kwargs.get("device_id", 0) is None:
            raise ValueError("ForwardPipeline needs a device handle")
        engine_kwargs = dict(engine_kwargs)
        engine_kwargs["streams"] = int(inner_streams)
        self.depth = int(depth)
        self.engines: List[Engine] = [Engine(**engine_kwargs) for _ in range(self.depth)]
        self.device = torch.device("cuda", self.engines[0].cfg.device_id)
        # One stream PRIORITY per slot, cycling through the device's range.  The runtime keeps a separate pool of hardware
        # queues per priority; same-priority streams share queues once a process holds more streams than queues (torch's own
        # pool is 32 per priority), and two slots that land on one queue run one after the other -- measured: 2488 instead of
        # 2820 images/s, depending on nothing but the order in which unrelated streams had been created
        # (profiles/r05_experiments.md).  Different priorities cannot collide.
        try:
            least, greatest = torch.cuda.Stream.priority_range()  # numerically least >= greatest (e.g. 0, -1)
        except Exception:
            least, greatest = 0, -1
        levels = list(range(least, greatest - 1, -1)) or [0]
        self.streams = [torch.cuda.Stream(device=self.device, priority=levels[i % len(levels)]) for i in range(self.depth)]
        self._levels = levels
        self._next = 0
        self.dual = bool(engine_kwargs.get("dual", False))
        # Does the overlap this class exists for actually happen?  (it silently depended on what the runtime did
        # with the streams.)  Neighbouring slots are probed with a sleeping one-wave kernel; a pair that serialises gets fresh
        # streams (other priority order first) up to `overlap_retries` times, and what is left is reported, not hidden.
        self.overlap_ratio: Optional[float] = None
        if check_overlap and self.depth >= 2:
            self.overlap_ratio = self.ensure_overlap(overlap_retries)

    SERIALISED = 1.5   # t(both) / t(alone) above this: the two streams did not run concurrently

    def measure_overlap(self) -> float:
        """Worst t(both) / t(alone) over neighbouring slot streams: ~1.0 concurrent, ~2.0 serialised."""
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        return max(probe_stream_overlap(self.streams[i], self.streams[(i + 1) % self.depth], idx) for i in range(self.depth - 1))

    def ensure_overlap(self, retries: int = 3) -> float:
        ratio = self.measure_overlap()
        for attempt in range(retries):
            if ratio < self.SERIALISED:
                break
            # new streams; the runtime hands out the least-referenced hardware queue of a priority level, so both a new stream
            # of the same level and the other level's pool are worth trying
            old = self.streams
            self.streams = [torch.cuda.Stream(device=self.device, priority=self._levels[(i + attempt + 1) % len(self._levels)])
                            for i in range(self.depth)]
            ratio = self.measure_overlap()
            del old
        if ratio >= self.SERIALISED:
            import warnings
            warnings.warn(f"omnidata_amd.ForwardPipeline: the slot streams do not run concurrently on this device (probe ratio "
                          f"{ratio:.2f}, 2.0 = fully serialised) -- forwards in flight will execute one after the other; "
                          f"throughput is that of single forwards.  (Too many streams alive in this process share the hardware queues.)")
        return ratio

    @classmethod
    def from_engine(cls, owner: Engine, depth: int = 2, inner_streams: int = 1, max_batch: Optional[int] = None) -> "ForwardPipeline":
        """`depth` fresh handles with `owner`'s configuration that all read `owner`'s weights (which must be on the device);
        `owner` itself stays free for ordinary forwards and is kept alive by the pipeline."""
        c = owner.cfg
        self = cls(depth=depth, inner_streams=inner_streams, num_channels=c.num_channels, max_batch=max_batch or c.max_batch,
                   dtype=owner.dtype, device_id=c.device_id, non_negative=bool(c.non_negative), ws_form=c.ws_form, ws_eps=c.ws_eps,
                   max_hw=(c.max_height, c.max_width), dual=bool(c.dual_task), x3_groups=c.x3_groups, backbone=owner.backbone,
                   flags=c.flags)
        for e in self.engines:
            e.share_weights_from(owner)
            e.copy_layer_precision_from(owner)   # per-layer precision is handle state, not part of the shared blob
            if owner.fp8_scales is not None:
                e.set_fp8_calibration(owner.fp8_scales)
        self._external_owner = owner
        return self

    # ---- weights: the first handle owns them, the others read the same blob
    def _share(self):
        for e in self.engines[1:]:
            e.share_weights_from(self.engines[0])

    def load_state_dict(self, sd):
        self.engines[0].load_state_dict(sd)
        self._share()

    def import_packed(self, blob: torch.Tensor):
        self.engines[0].import_packed(blob)
        self._share()

    def export_packed(self) -> torch.Tensor:
        return self.engines[0].export_packed()

    @property
    def packed_bytes(self) -> int:
        return self.engines[0].packed_bytes

    def calibrate_fp8(self, x: torch.Tensor):
        """dtype 'fp8': the first handle measures the activation scales, every handle installs the same ones."""
        self.engines[0].calibrate_fp8(x)
        s, _ = self.engines[0].fp8_calibration()
        for e in self.engines[1:]:
            e.set_fp8_calibration(s)

    # ---- forwards
    def submit(self, x: torch.Tensor, out=None, out_depth=None) -> Ticket:
        """Enqueues one forward of `x` ([B,3,H,W], B <= max_batch) on the next handle.  `x` (and `out`) must be valid in the
        CURRENT stream's order at the time of the call; the forward starts once the current stream has reached this point AND
        the handle's previous forward is done -- not when earlier tickets have been waited for, which is what lets it overlap
        the forward submitted just before.  dual handles: `out` = normals, `out_depth` = depth; the ticket's `out` is the pair."""
        slot = self._next
        self._next = (self._next + 1) % self.depth
        eng, st = self.engines[slot], self.streams[slot]
        cur = torch.cuda.current_stream(self.device)
        # conversions and output allocations happen on the CALLER's stream (its allocator pool): only the forward itself runs
        # on the slot's stream
        if x.is_cuda and x.dim() == 4:
            if x.dtype not in IO_DTYPES:
                x = x.float()
            x = x.contiguous()
            B, _, H, W = x.shape
            if self.dual:
                if out is None:
                    out = torch.empty(B, 3, H, W, dtype=x.dtype, device=x.device)
                if out_depth is None:
                    out_depth = torch.empty(B, 1, H, W, dtype=x.dtype, device=x.device)
            elif out is None:
                out = torch.empty(B, eng.cfg.num_channels, H, W, dtype=x.dtype, device=x.device)
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            if self.dual:
                res = eng.forward_dual(x, out_normal=out, out_depth=out_depth)
            else:
                res = eng.forward(x, out=out)
            ev = torch.cuda.Event()
            ev.record(st)
        # the caching allocator must not hand these blocks to another stream's tensor before the forward has run
        x.record_stream(st)
        for t in (res if isinstance(res, tuple) else (res,)):
            t.record_stream(st)
        return Ticket(res, ev)

    def forward(self, x: torch.Tensor, out=None):
        """Stream-ordered convenience (the semantics of Engine.forward): submit + wait.  No overlap between calls."""
        return self.submit(x, out=out).wait()

    def map(self, batches: Iterable[torch.Tensor]) -> Iterator:
        """Results of `batches` in order with `depth` forwards in flight; each result is ordered into the current stream
        before it is yielded."""
        pending: List[Ticket] = []
        for x in batches:
            pending.append(self.submit(x))
            if len(pending) >= self.depth:
                yield pending.pop(0).wait()
        while pending:
            yield pending.pop(0).wait()

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def close(self):
        for e in reversed(self.engines):   # sharers before the owner
            e.close()
        self.engines = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
