"""Host-side mirror of the reference's model class for the one supported path.

``DPTDepthModel`` keeps the constructor, ``state_dict`` key names, ``load_state_dict`` /
``.to()`` / ``.eval()`` behaviour and the forward contract of
``omnidata_tools/torch/modules/midas/dpt_depth.py:87-107`` (DPT.forward :67-85), but the
forward pass runs in libdptx.so (hand-written gfx950 kernels) instead of ATen.
PyTorch is only plumbing here: parameter storage, device memory, streams.

There is deliberately NO CPU / eager fallback: calling the model with CPU tensors, without a
GPU, or without the built extension raises.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from .engine import Engine
from .weights import (compose_dual_state_dict, dual_state_dict_spec, random_dual_state_dict, random_state_dict,
                      read_checkpoint, state_dict_spec, BACKBONES)


def _io_dtype(x: torch.Tensor) -> torch.dtype:
    """Result element type: the input's when the engine reads it directly (fp32 / bf16 / fp16), else fp32."""
    return x.dtype if x.dtype in (torch.float32, torch.bfloat16, torch.float16) else torch.float32


class _Node(nn.Module):
    """Anonymous container so that nested parameter names equal the reference's keys."""


class BaseModel(nn.Module):
    def load(self, path: str):
        """Same contract as modules/midas/base_model.py:4-16 (plus Lightning .ckpt dicts)."""
        self.load_state_dict(read_checkpoint(path))


class _EngineGuards:
    """What DPTDepthModel and DPTDualTaskModel share around an engine call: the fp16 range guard and the fp8 calibration.

    Range guard (``overflow_fallback``, default on).  The fp16-plane modes ('mixed', 'fp16x3', 'fp16') cannot represent
    |x| > 65504 (two planes: ~1.3e5) and nothing inside the forward clamps; the parity claims are validated on synthetic
    weight families only.  Every forward of such an engine scans the first head convolution's output on the device --
    the tensor every decoder path and, through them, the ViT blocks reach by residual additions -- and ORs "non-finite"
    into a sticky device flag (include/dptx.h dptx_range_status; ~0.3 % of a forward).  The model reads the flag after
    the FIRST forward of a set of weights and after every ``range_check_every``-th one (one stream synchronisation
    each): when it is set although the input was finite, the model switches to the bf16-plane mode of the same kind
    ('bf16x3' for the parity modes, 'bf16' for 'fp16': fp32's exponent range) with a warning and recomputes the
    current batch; between two reads at most ``range_check_every`` - 1 earlier results can be affected, which the
    warning says.  The activations are checked, not the result: the decoder's ReLUs turn a NaN back into 0, so an
    overflow upstream can leave a finite, wrong output.

    fp8 calibration.  The per-tensor activation scales of the e4m3 copies are data: ``calibrate(x)`` measures them on a
    representative batch.  A model that was never calibrated does it on the first batch it sees and says so (warning):
    replicated ranks must not each calibrate on their own shard (``dist.broadcast_fp8_calibration``).  The scales
    belong to the weights, not to the engine object: they are re-installed when the engine is rebuilt (``.to()``, a
    larger input size) and dropped when new weights are loaded."""

    _FP16_FALLBACK = {"mixed": "bf16x3", "fp16x3": "bf16x3", "fp16": "bf16"}
    range_check_every = 16

    def _init_guards(self, overflow_fallback: bool):
        self.overflow_fallback = bool(overflow_fallback)
        self._range_checked = None   # (values version, dtype) whose first forward came back clean
        self._since_range_check = 0
        self._values_version = 0     # bumped by load_state_dict only (``.to()`` moves the same values)
        self._fp8_scales = None      # (values version, scales) of the last calibration

    def _restore_fp8(self, eng):
        """A rebuilt engine gets the scales of the same weights back."""
        if self.engine_dtype == "fp8" and not eng.fp8_calibrated and self._fp8_scales is not None \
                and self._fp8_scales[0] == self._values_version:
            eng.set_fp8_calibration(self._fp8_scales[1])

    def _ensure_fp8(self, eng, x):
        if self.engine_dtype != "fp8":
            return
        self._restore_fp8(eng)
        if not eng.fp8_calibrated:
            import warnings
            warnings.warn("omnidata_amd: dtype='fp8' model was not calibrated -- measuring the activation scales of the e4m3 "
                          "tensors on this first batch.  Call model.calibrate(x) with a representative batch (and "
                          "omnidata_amd.dist.broadcast_fp8_calibration across ranks) to choose them explicitly.")
            eng.calibrate_fp8(x)
        if eng.fp8_scales is not None:
            self._fp8_scales = (self._values_version, eng.fp8_scales)

    @torch.no_grad()
    def calibrate(self, x: torch.Tensor):
        """dtype='fp8' only: (re-)measures the activation ranges of the e4m3 tensors on `x` (at most max_batch images)."""
        if self.engine_dtype != "fp8":
            raise RuntimeError("calibrate() applies to dtype='fp8'")
        eng = self._get_engine(x.device)
        eng.calibrate_fp8(x[: self._chunk()])
        self._fp8_scales = (self._values_version, eng.fp8_scales)

    def _range_fallback_needed(self, eng, x: torch.Tensor, rerun=None, engines=None, pre_sync=None) -> bool:
        """Called after a forward.  `engines` (forward_pipelined): every handle whose sticky flag belongs to this model's
        forwards -- all of them are read and reset, after `pre_sync()` has drained their streams.  True: the engine left the fp16 range, the model has switched to bf16 planes and the
        caller recomputes the batch.  `rerun()` repeats the forward of THIS batch on the same engine: the device flag is
        sticky over up to `range_check_every` forwards, so a periodic read that finds it set cannot tell the arithmetic
        from a NaN / Inf INPUT image in an earlier batch -- the current, finite batch is run once more on the
        reset flag and the dtype is only left if that forward sets it again."""
        if not (self.overflow_fallback and self.engine_dtype in self._FP16_FALLBACK):
            return False
        tag = (self._values_version, self.engine_dtype)
        first = self._range_checked != tag
        self._since_range_check += 1
        if not first and self._since_range_check < self.range_check_every:
            return False
        n_since, self._since_range_check = self._since_range_check, 0

        def overflowed() -> bool:
            if pre_sync is not None:
                pre_sync()
            return any([e.range_overflowed(reset=True) for e in (engines or [eng])])   # a list: every flag is reset
        if not overflowed():
            self._range_checked = tag
            return False
        if not bool(torch.isfinite(x).all()):
            return False  # the input's problem, not the arithmetic's
        import warnings
        if not first and rerun is not None:
            rerun()
            if not overflowed():
                self._range_checked = tag
                warnings.warn(f"omnidata_amd: dtype={self.engine_dtype!r} saw non-finite activations in one of the previous "
                              f"{n_since - 1} forwards, but not on this (finite) batch run again: most likely a NaN / Inf input "
                              f"image in an earlier batch; keeping the dtype.  Results of those forwards may be non-finite.")
                return False
        safe = self._FP16_FALLBACK[self.engine_dtype]
        warnings.warn(f"omnidata_amd: dtype={self.engine_dtype!r} produced non-finite activations -- a tensor exceeds the fp16 "
                      f"range (65504) with these weights; switching this model to dtype={safe!r} (bf16 planes: fp32's range) and "
                      f"recomputing this batch" + ("" if first else f"; up to {n_since - 1} earlier results since the last check "
                      f"may be affected") + ".  Pass overflow_fallback=False to keep the dtype.")
        self.engine_dtype = safe
        self.x3_groups = 0
        return True


class DPTDepthModel(_EngineGuards, BaseModel):
    """Drop-in for ``DPTDepthModel(backbone={'vitb_rn50_384' | 'vitl16_384'}, num_channels={1,3})``
    (DPT-Hybrid, the published omnidata configuration, and DPT-Large, demo.py:81 / dpt_depth.py:41-45).

    Extra keyword arguments (engine side): ``dtype`` in {'mixed','fp16x3','bf16x3','fp16','bf16','fp8'} -- MFMA
    operand / activation storage type.  The DEFAULT is 'mixed', the mode that matches the reference: the layer groups
    in ``x3_groups`` ('resnet+embed+reassemble+rn+fusion+head' by default: everything but the ViT blocks) keep hi/lo
    fp16 planes and spend 3 MFMAs per product, the ViT blocks run single-pass fp16 -- within north_star's 1e-3 of the
    fp32 reference forward (profiles/r02_precision_frontier.md, oracle/precision_layers.py).  'fp16x3' / 'bf16x3' run
    everything with 3 MFMAs (reference-grade, ~1e-5 / ~1e-4).  'bf16' / 'fp16' are single-pass THROUGHPUT modes: ~1.6x
    faster, but ~6e-2 / ~9e-3 max-abs from the reference on the seeded weights -- NOT within 1e-3; 'fp8' additionally
    runs decoder convolutions on e4m3 operands: by default the six that keep it within 2x the bf16 mode's error
    (oracle/fp8_layers.py) -- measured within 2 % of 'bf16', i.e. NOT a throughput mode (rounds 5-6) -- with ``fp8_all=True``
    all 19 eligible ones (+9 %, 7-9 degrees of mean angular error).  ``max_batch`` -- arena size (larger batches are chunked).

    ``overflow_fallback`` (default on): the fp16 range guard described in ``_EngineGuards`` -- an on-device scan per
    forward, read after the first forward of a set of weights and every 16th one afterwards; falls back to bf16 planes.
    """

    def __init__(self, path: Optional[str] = None, non_negative: bool = True, num_channels: int = 1,
                 backbone: str = "vitb_rn50_384", features: int = 256, readout: str = "project",
                 channels_last: bool = False, use_bn: bool = False, dtype: str = "mixed",
                 max_batch: int = 32, init_seed: int = 0, x3_groups=0, overflow_fallback: bool = True, fp8_all: bool = False,
                 fp8_vit: bool = False):
        super().__init__()
        self.fp8_all = bool(fp8_all)   # dtype 'fp8': all 19 eligible decoder convs on e4m3 (lossy) instead of the six safe ones
        self.fp8_vit = bool(fp8_vit)   # dtype 'fp8': qkv / fc1 / fc2 of the ViT blocks on e4m3 too (include/dptx.h DPTX_FLAG_FP8_VIT)
        if backbone not in BACKBONES:
            # blocks.py:42-44: unknown backbones print and assert
            print(f"Backbone '{backbone}' not implemented")
            assert False, f"backbones built for MI355X: {BACKBONES} (DPT-Hybrid, DPT-Large)"
        if features != 256 or readout != "project" or use_bn:
            raise NotImplementedError("only features=256, readout='project', use_bn=False (the published "
                                      "omnidata DPT-Hybrid configuration) is supported")
        if num_channels not in (1, 3):
            raise ValueError("num_channels must be 1 (depth) or 3 (surface normals)")
        self.num_channels = num_channels
        self.backbone = backbone
        self.non_negative = bool(non_negative)
        self.channels_last = channels_last  # accepted and, as in the reference (dpt_depth.py:68-69), a no-op
        self.engine_dtype = dtype
        self.x3_groups = x3_groups
        self._init_guards(overflow_fallback)
        self.max_batch = max(1, min(int(max_batch), 48))  # engine limit; larger batches are chunked in forward()
        self.max_hw = (384, 384)  # arena is planned for this input size; grows on demand (forward_flex, vit.py:119)
        init = random_state_dict(init_seed, num_channels, backbone=backbone)
        for key, shape in state_dict_spec(num_channels, backbone=backbone).items():
            *mods, leaf = key.split(".")
            node = self
            for m in mods:
                if not hasattr(node, m):
                    node.add_module(m, _Node())
                node = getattr(node, m)
            node.register_parameter(leaf, nn.Parameter(init[key], requires_grad=False))
        self._engine: Optional[Engine] = None
        self._engine_key = None
        self._pipe, self._pipe_key = None, None   # forward_pipelined's cached ForwardPipeline
        self._weights_version = 0
        if path is not None:
            self.load(path)

    # ---- keep the engine in sync with the parameters
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self._weights_version += 1
        self._values_version += 1
        return r

    @torch.no_grad()
    def forward_pipelined(self, batches, depth: int = 2):
        """Throughput form of ``model(x)`` for a STREAM of batches (each [B,3,H,W], any B -- batches larger than ``max_batch`` are
        chunked as in ``forward`` --, 384x384 or any one size the engine is planned for): a generator of the results in order,
        with ``depth`` whole-batch forwards in flight on the GPU (omnidata_amd/pipeline.py; each result is bit-identical to
        ``model(x)``).  The pipeline (``depth`` arenas next to the model's own engine) is cached on the model and rebuilt only
        when the engine is (new weights, ``.to()``, another dtype, a larger input).  The fp16 range guard runs at ``forward``'s
        cadence -- on the first batch of a set of weights and every ``range_check_every``-th one the pipeline is drained and every
        handle's flag is read BEFORE the batch is yielded; on overflow the model switches to bf16 planes exactly as ``forward``
        does and the batches still in flight are recomputed through ``forward`` (results already yielded since the previous
        check may be affected, which the warning says)."""
        pending = []   # (x, y, tickets) in submission order

        def finish(entry):
            for t in entry[2]:
                t.wait()
            return entry[1]

        def retire(entry):
            """-> list of results to yield for `entry` (and, after a dtype switch, for everything that was still in flight)."""
            x, y, _ = entry
            pipe = self._pipe
            finish(entry)

            def rerun():   # the same batch once more on one handle, on the current stream
                step = self._chunk()
                for i in range(0, x.shape[0], step):
                    pipe.engines[0].forward(x[i:i + step], out=y[i:i + step])
            if self._range_fallback_needed(pipe.engines[0], x, rerun, engines=pipe.engines, pre_sync=pipe.synchronize):
                redo = [entry] + pending[:]
                del pending[:]
                self._drop_pipeline()
                return [self.forward(e[0]) for e in redo]   # on the bf16-plane engine (forward squeezes)
            return [y.squeeze(dim=1)]

        try:
            for x in batches:
                if not x.is_cuda:
                    raise RuntimeError("omnidata_amd.DPTDepthModel runs only on an AMD GPU (HIP); there is no CPU fallback")
                if x.dim() != 4 or x.shape[2] % 32 or x.shape[3] % 32:
                    raise ValueError(f"expected [B,3,H,W] with H, W multiples of 32, got {tuple(x.shape)}")
                B, _, H, W = x.shape
                if H * W > self.max_hw[0] * self.max_hw[1]:
                    for e in pending:     # the arena is re-planned: finish what runs on the old one first
                        finish(e)
                    self.max_hw = (H, W)
                pipe = self._get_pipeline(x.device, depth, x)
                step = self._chunk()
                y = torch.empty(B, self.num_channels, H, W, dtype=_io_dtype(x), device=x.device)
                pending.append((x, y, [pipe.submit(x[i:i + step], out=y[i:i + step]) for i in range(0, B, step)]))
                if len(pending) >= depth:
                    for r in retire(pending.pop(0)):
                        yield r
            while pending:
                for r in retire(pending.pop(0)):
                    yield r
        finally:
            if self._pipe is not None:
                self._pipe.synchronize()

    def _get_pipeline(self, device: torch.device, depth: int, x: Optional[torch.Tensor] = None):
        """The cached ForwardPipeline over the model's engine (not one per call -- `depth` arenas of 5.2 GB at
        B = 32, a hipMalloc / hipFree and stream creation each time)."""
        from .pipeline import ForwardPipeline
        eng = self._get_engine(device)
        if x is not None:
            self._ensure_fp8(eng, x[: self._chunk()])
        key = (self._engine_key, int(depth), tuple(sorted(eng.layer_precision.items())),
               None if eng.fp8_scales is None else eng.fp8_scales.tobytes())
        if self._pipe is None or self._pipe_key != key or self._pipe._external_owner is not eng:
            self._drop_pipeline()
            self._pipe, self._pipe_key = ForwardPipeline.from_engine(eng, depth=depth), key
        return self._pipe

    def _drop_pipeline(self):
        if getattr(self, "_pipe", None) is not None:
            self._pipe.synchronize()
            self._pipe.close()
        self._pipe, self._pipe_key = None, None

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self._weights_version += 1
        return r

    def _get_engine(self, device: torch.device) -> Engine:
        key = (device.index if device.index is not None else torch.cuda.current_device(),
               self._weights_version, self.engine_dtype, self._chunk(), self.max_hw)
        if self._engine is None or self._engine_key != key:
            self._drop_pipeline()   # its handles read the old engine's weights and arena plan
            if self._engine is not None:
                self._engine.close()
            eng = Engine(num_channels=self.num_channels, max_batch=self._chunk(), dtype=self.engine_dtype,
                         device_id=key[0], non_negative=self.non_negative, max_hw=self.max_hw,
                         x3_groups=self.x3_groups, backbone=self.backbone,
                         flags=(16 if self.fp8_all else 0) | (32 if self.fp8_vit else 0))
            eng.load_state_dict(super().state_dict())
            self._engine, self._engine_key = eng, key
        return self._engine

    def _chunk(self) -> int:
        """Images per engine call: every activation must stay below 2 GB (32-bit buffer offsets, include/dptx.h)."""
        return max(1, min(self.max_batch, ((1 << 31) - 1) // (self.max_hw[0] * self.max_hw[1] * 256)))

    @property
    def engine(self) -> Optional[Engine]:
        return self._engine

    def adopt_engine(self, engine: Engine, device_index: int):
        """Use an engine whose packed weights arrived by broadcast (multi-GPU start-up)."""
        self._engine = engine
        self._engine_key = (device_index, self._weights_version, self.engine_dtype, self._chunk(), self.max_hw)

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("omnidata_amd.DPTDepthModel runs only on an AMD GPU (HIP); got a CPU tensor. "
                               "There is no CPU fallback -- use the reference implementation on CPU.")
        if x.dim() != 4 or x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError(f"expected [B,3,H,W] with H, W multiples of 32, got {tuple(x.shape)}")
        B, _, H, W = x.shape
        if H * W > self.max_hw[0] * self.max_hw[1]:
            self.max_hw = (H, W)  # re-plans the arena (and re-packs the weights) once for the larger size
        eng = self._get_engine(x.device)
        step = self._chunk()
        self._ensure_fp8(eng, x[:step])
        y = torch.empty(B, self.num_channels, H, W, dtype=_io_dtype(x), device=x.device)

        def run():
            for i in range(0, B, step):
                eng.forward(x[i:i + step], out=y[i:i + step])
        run()
        if self._range_fallback_needed(eng, x, run):
            return self.forward(x)  # on the bf16-plane engine
        return y.squeeze(dim=1)  # dpt_depth.py:106-107


class DPTDualTaskModel(_EngineGuards, nn.Module):
    """Surface normals AND depth from one encoder pass (BASELINE.json configs[4]).

    ``pretrained.*`` (ResNetV2-50 + ViT-B + read-outs) runs once; ``scratch.*`` (normal decoder, 3 channels) and
    ``depth.scratch.*`` (depth decoder, 1 channel) both consume its four feature maps: 185.3 GMAC per image
    instead of 2 x 127.6.  The reference has no such model -- its two checkpoints are independently fine-tuned
    full networks -- so this is exact only for weights that share ``pretrained.*``; ``from_single_task`` picks
    which checkpoint donates the encoder.  Both heads see the same input tensor ([0,1] convention).

    forward(x [B,3,H,W]) -> (normal [B,3,H,W], depth [B,H,W]).
    """

    def __init__(self, dtype: str = "mixed", max_batch: int = 32, init_seed: int = 0, non_negative: bool = True,
                 x3_groups=0, overflow_fallback: bool = True, fp8_all: bool = False, fp8_vit: bool = False):
        super().__init__()
        self._init_guards(overflow_fallback)
        self.fp8_all = bool(fp8_all)
        self.fp8_vit = bool(fp8_vit)
        self.engine_dtype = dtype
        self.x3_groups = x3_groups
        self.max_batch = max(1, min(int(max_batch), 48))
        self.max_hw = (384, 384)
        self.non_negative = bool(non_negative)
        init = random_dual_state_dict(init_seed)
        for key in dual_state_dict_spec():
            *mods, leaf = key.split(".")
            node = self
            for m in mods:
                if not hasattr(node, m):
                    node.add_module(m, _Node())
                node = getattr(node, m)
            node.register_parameter(leaf, nn.Parameter(init[key], requires_grad=False))
        self._engine: Optional[Engine] = None
        self._engine_key = None
        self._weights_version = 0

    @classmethod
    def from_single_task(cls, normal_sd: Dict[str, torch.Tensor], depth_sd: Dict[str, torch.Tensor],
                         backbone: str = "normal", **kw) -> "DPTDualTaskModel":
        m = cls(**kw)
        m.load_state_dict(compose_dual_state_dict(normal_sd, depth_sd, backbone))
        return m

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self._weights_version += 1
        self._values_version += 1
        return r

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self._weights_version += 1
        return r

    def _chunk(self) -> int:
        return max(1, min(self.max_batch, ((1 << 31) - 1) // (self.max_hw[0] * self.max_hw[1] * 256)))

    @property
    def engine(self) -> Optional[Engine]:
        return self._engine

    def _get_engine(self, device: torch.device) -> Engine:
        key = (device.index if device.index is not None else torch.cuda.current_device(),
               self._weights_version, self.engine_dtype, self._chunk(), self.max_hw)
        if self._engine is None or self._engine_key != key:
            if self._engine is not None:
                self._engine.close()
            eng = Engine(num_channels=3, max_batch=self._chunk(), dtype=self.engine_dtype, device_id=key[0],
                         non_negative=self.non_negative, max_hw=self.max_hw, dual=True, x3_groups=self.x3_groups,
                         flags=(16 if self.fp8_all else 0) | (32 if self.fp8_vit else 0))
            eng.load_state_dict(super().state_dict())
            self._engine, self._engine_key = eng, key
        return self._engine

    @torch.no_grad()
    def forward(self, x: torch.Tensor):
        if not x.is_cuda:
            raise RuntimeError("omnidata_amd.DPTDualTaskModel runs only on an AMD GPU (HIP); there is no CPU fallback")
        if x.dim() != 4 or x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError(f"expected [B,3,H,W] with H, W multiples of 32, got {tuple(x.shape)}")
        B, _, H, W = x.shape
        if H * W > self.max_hw[0] * self.max_hw[1]:
            self.max_hw = (H, W)
        eng = self._get_engine(x.device)
        step = self._chunk()
        self._ensure_fp8(eng, x[:step])
        yn = torch.empty(B, 3, H, W, dtype=_io_dtype(x), device=x.device)
        yd = torch.empty(B, 1, H, W, dtype=_io_dtype(x), device=x.device)
        def run():
            for i in range(0, B, step):
                eng.forward_dual(x[i:i + step], out_normal=yn[i:i + step], out_depth=yd[i:i + step])
        run()
        if self._range_fallback_needed(eng, x, run):
            return self.forward(x)  # on the bf16-plane engine
        return yn, yd.squeeze(dim=1)


def build_model(task: str = "normal", weights: Optional[str] = None, random_weights: Optional[int] = None,
                backbone: str = "vitb_rn50_384", **kw) -> DPTDepthModel:
    """normal -> 3 channels, depth -> 1 channel (demo.py:63,82); backbone 'vitl16_384' = DPT-Large (demo.py:81)."""
    if task not in ("normal", "depth"):
        raise ValueError("task should be one of the following: normal, depth")
    C = 3 if task == "normal" else 1
    model = DPTDepthModel(backbone=backbone, num_channels=C,
                          init_seed=0 if random_weights is None else random_weights, **kw)
    if weights is not None:
        model.load_state_dict(read_checkpoint(weights))
    return model.eval()
