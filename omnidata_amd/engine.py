"""ctypes binding of libdptx.so (include/dptx.h).  No CPU fallback: if the library or a GPU is
missing the compute entry points raise -- the product never routes through oracle/."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DPTX_LIB") or os.path.join(_HERE, "libdptx.so")  # $DPTX_LIB: an experiment build (build.py)

DTYPES = {"bf16": 0, "fp16": 1, "bf16x3": 2, "fp16x3": 3, "mixed": 4, "fp8": 5}
# layer groups of dptx_config.x3_groups (include/dptx.h DPTX_GROUP_*)
GROUPS = {"resnet": 1, "embed": 2, "vit": 4, "reassemble": 8, "rn": 16, "fusion": 32, "head": 64}


def groups_mask(names) -> int:
    """'resnet+embed' / ['resnet', 'embed'] / int -> DPTX_GROUP_* bit mask."""
    if isinstance(names, int):
        return names
    if isinstance(names, str):
        names = [n for n in names.replace(",", "+").split("+") if n]
    return sum(GROUPS[n] for n in set(names))
# include/dptx.h DPTX_BACKBONE_*
BACKBONE_IDS = {"vitb_rn50_384": 0, "vitl16_384": 1}
ERRORS = {0: "ok", -1: "invalid argument / call order", -2: "state_dict key error", -3: "HIP error",
          -4: "no device", -5: "allocation failed"}


class DptxConfig(C.Structure):
    _fields_ = [("num_channels", C.c_int32), ("max_batch", C.c_int32), ("dtype", C.c_int32),
                ("device_id", C.c_int32), ("non_negative", C.c_int32), ("ws_form", C.c_int32),
                ("ws_eps", C.c_float), ("max_height", C.c_int32), ("max_width", C.c_int32),
                ("dual_task", C.c_int32), ("streams", C.c_int32), ("x3_groups", C.c_int32), ("backbone", C.c_int32),
                ("flags", C.c_int32), ("reserved", C.c_int32)]


# (name, restype, argtypes) for every symbol declared in include/dptx.h
_vp, _i32, _i64p, _f32p, _sz = C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_float), C.c_size_t
ABI = [
    ("dptx_default_config", None, [C.POINTER(DptxConfig)]),
    ("dptx_create", C.c_int, [C.POINTER(_vp), C.POINTER(DptxConfig)]),
    ("dptx_destroy", None, [_vp]),
    ("dptx_load_tensor", C.c_int, [_vp, C.c_char_p, _vp, _i64p, _i32]),
    ("dptx_finalize_weights", C.c_int, [_vp]),
    ("dptx_packed_bytes", _sz, [_vp]),
    ("dptx_export_packed_host", C.c_int, [_vp, _vp, _sz]),
    ("dptx_export_packed_device", C.c_int, [_vp, _vp, _sz, _vp]),
    ("dptx_import_packed_device", C.c_int, [_vp, _vp, _sz, _vp]),
    ("dptx_workspace_bytes", _sz, [_vp]),
    ("dptx_forward", C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp]),
    ("dptx_forward_hw", C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _i32, _vp]),
    ("dptx_forward_dual", C.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp]),
    ("dptx_set_layer_precision", C.c_int, [_vp, C.c_char_p, _i32]),
    ("dptx_tune_schedule", C.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    ("dptx_schedule_info", C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _f32p, _f32p]),
    ("dptx_probe_stream_overlap", C.c_int, [_i32, _vp, _vp, _i32, _f32p]),
    ("dptx_calibrate_fp8", C.c_int, [_vp, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _vp]),
    ("dptx_fp8_get_calibration", C.c_int, [_vp, _f32p, _f32p, _i32]),
    ("dptx_share_packed", C.c_int, [_vp, _vp]),
    ("dptx_fp8_set_calibration", C.c_int, [_vp, _f32p, _i32]),
    ("dptx_range_status", C.c_int, [_vp, C.POINTER(C.c_int32), _i32, _vp]),
    ("dptx_tap", C.c_int, [_vp, C.c_char_p, _vp, _sz, _i64p]),
    ("dptx_enable_taps", C.c_int, [_vp, C.c_int]),
    ("dptx_forward_info", C.c_int, [_vp, _i64p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("dptx_set_profiling", C.c_int, [_vp, C.c_int]),
    ("dptx_profile_get", C.c_int, [_vp, _i32, C.POINTER(C.c_double), _i64p, C.POINTER(C.c_double)]),
    ("dptx_profile_dump", C.c_int, [_vp, C.c_char_p]),
    ("dptx_last_error", C.c_char_p, [_vp]),
    ("dptx_version", C.c_char_p, []),
    ("dptx_preprocess_u8", C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    ("dptx_postprocess_normal_u8", C.c_int, [_vp, _vp, _vp]),
    ("dptx_postprocess_depth", C.c_int, [_vp, _vp, _vp]),
    ("dptx_resample_coeffs", C.c_int, [_i32, _i32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _i32, C.POINTER(C.c_int32)]),
    ("dptx_op_set_planes", C.c_int, [C.c_int64, C.c_int64]),
    ("dptx_op_gemm", C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp]),
    ("dptx_op_conv", C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp] + [_i32] * 13 + [_vp]),
    ("dptx_op_conv_planes", C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp] + [_i32] * 16 + [_vp]),
    ("dptx_op_stem_conv", C.c_int, [_i32, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    ("dptx_op_attention", C.c_int, [_i32, _vp, _vp, _i32, _i32, _i32, _vp]),
    ("dptx_op_layernorm", C.c_int, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, C.c_float, _vp]),
    ("dptx_op_groupnorm", C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, C.c_float, _vp, _vp]),
    ("dptx_debug_set_trace", C.c_int, [_vp]),
    ("dptx_debug_set_gemm_flags", C.c_int, [C.c_int32]),
    ("dptx_debug_arena_fill", C.c_int, [_vp, _i32]),
    ("dptx_debug_arena_read", C.c_int, [_vp, _vp, _sz, _sz]),
    ("dptx_debug_arena_layout", C.c_int, [_vp, C.c_char_p, _sz]),
    ("dptx_debug_arena_checksums", C.c_int, [_vp, _vp, _i32, _vp]),
    ("dptx_debug_set_launch_sums", C.c_int, [_vp, _vp, _i32]),
    ("dptx_op_conv_fp8", C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp] + [_i32] * 13 + [C.c_float, _vp]),
    ("dptx_op_conv_groupnorm", C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp] + [_i32] * 12 + [C.c_float, _vp, _vp]),
    ("dptx_op_upsample2x", C.c_int, [_i32, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    ("dptx_op_gemm_ln", C.c_int, [_i32, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _i32, C.c_float, _vp]),
    ("dptx_op_gemm_stream", C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    ("dptx_op_gemm_stream32", C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    ("dptx_op_head_tail", C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
]

_lib = None


def _embedded_hash(path: str) -> str:
    """`src=<hash>` of a built library, read from the FILE's bytes (the dptx_version() literal "dptx ... src=<16 hex>").
    Not by loading it: glibc matches an already-mapped object by name before it looks at the file and ctypes never
    dlcloses, so a stale copy mapped here would be what a later CDLL(path) returns even after build() replaced the file
    (the first run after any source change then died on the stale mapping)."""
    import re
    try:
        with open(path, "rb") as f:
            data = f.read()
    except OSError:
        return ""
    m = re.search(rb"dptx [0-9.]+ \(gfx950[^)]*\) src=([0-9a-f]{16})", data)
    return m.group(1).decode() if m else ""


def load_library() -> C.CDLL:
    """Loads libdptx.so and declares every prototype.  The binary must have been built from the sources it sits next to
    (the .so travels with the tree; mtimes and side files prove nothing on another machine): build.py embeds the sha256 of
    csrc/* + include/dptx.h in the library (dptx_version() ends in "src=<hash>") and THAT is what is compared with the tree
    here -- a correct prebuilt library needs neither hipcc nor the build directory.  A missing or stale library is rebuilt
    when hipcc is available (it cross-compiles without a GPU), by one process at a time (build.py takes a file lock and
    links to a temporary file that is renamed into place, so ranks started together do not write one object directory at
    once); otherwise this raises.  There is no CPU fallback."""
    global _lib
    if _lib is None:
        check = not os.environ.get("DPTX_LIB") and not os.environ.get("DPTX_SKIP_HASH_CHECK")
        want = ""
        if check:
            from .build import build, source_hash
            want = source_hash(os.environ.get("DPTX_CXXFLAGS", "").split())
            have = _embedded_hash(LIB_PATH) if os.path.exists(LIB_PATH) else ""
            if have != want:
                hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
                if not os.path.exists(hipcc):
                    raise RuntimeError(f"{LIB_PATH} is {'stale (built from sources ' + have + ')' if have else 'missing'}; the tree "
                                       f"has {want} and {hipcc} is not available: run `python -m omnidata_amd.build` on a "
                                       "machine with ROCm. There is no CPU fallback.")
                build()  # nothing of the stale file is mapped in this process: its hash was read from the bytes
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} not found: run `python -m omnidata_amd.build` "
                               "(or __graft_entry__.build()). There is no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, res, args in ABI:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        if check:
            built = lib.dptx_version().decode().rsplit("src=", 1)[-1]
            if built != want:
                raise RuntimeError(f"{LIB_PATH} is stale: built from sources {built}, the tree has {want}. "
                                   "Run `python -m omnidata_amd.build --force`.")
        _lib = lib
    return _lib


def _ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def _stream(device=None) -> int:
    """hipStream_t of torch's current stream on `device` (default: the current device)."""
    return torch.cuda.current_stream(device).cuda_stream


IO_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}   # include/dptx.h DPTX_IO_*


def probe_stream_overlap(stream_a, stream_b, device_index: int = 0, spin_us: int = 2000) -> float:
    """t(both streams busy) / t(one stream busy) for a sleeping one-wave kernel (include/dptx.h dptx_probe_stream_overlap):
    ~1.0 = the two torch streams run concurrently, ~2.0 = the runtime serialises them (one hardware queue).  Blocks."""
    r = C.c_float(0)
    rc = load_library().dptx_probe_stream_overlap(int(device_index), stream_a.cuda_stream, stream_b.cuda_stream, int(spin_us), C.byref(r))
    if rc != 0:
        raise RuntimeError(f"dptx_probe_stream_overlap failed: {ERRORS.get(rc, rc)}")
    return float(r.value)


def _check_out(out: torch.Tensor, shape, device, dtype=torch.float32):
    if (out.dtype != dtype or not out.is_contiguous() or out.device != device or tuple(out.shape) != tuple(shape)):
        raise ValueError(f"out must be a contiguous {dtype} tensor of shape {tuple(shape)} on {device}, got "
                         f"{out.dtype} {tuple(out.shape)} on {out.device} (contiguous={out.is_contiguous()})")


class Engine:
    """One dptx handle: packed weights + activation arena on one GPU (or host-only packing)."""

    def __init__(self, num_channels: int = 3, max_batch: int = 32, dtype: str = "mixed",
                 device_id: Optional[int] = 0, non_negative: bool = True, ws_form: int = 0, ws_eps: float = 1e-8,
                 max_hw: Tuple[int, int] = (384, 384), dual: bool = False, streams: int = 0, x3_groups=0,
                 backbone: str = "vitb_rn50_384", flags: int = 0):
        self.lib = load_library()
        cfg = DptxConfig()
        self.lib.dptx_default_config(C.byref(cfg))
        cfg.num_channels, cfg.max_batch, cfg.dtype = num_channels, max_batch, DTYPES[dtype]
        cfg.device_id = -1 if device_id is None else int(device_id)
        cfg.non_negative, cfg.ws_form, cfg.ws_eps = int(non_negative), ws_form, ws_eps
        cfg.max_height, cfg.max_width = int(max_hw[0]), int(max_hw[1])
        cfg.dual_task = int(dual)
        cfg.streams = int(streams)
        cfg.x3_groups = groups_mask(x3_groups)  # dtype "mixed": groups that run 3 MFMAs per product (0 = all but the ViT blocks)
        cfg.backbone = BACKBONE_IDS[backbone]
        cfg.flags = int(flags)  # include/dptx.h DPTX_FLAG_* (1: no LayerNorm fold, 2: group-level precision policy only, 4: fp32 token stream in the single-pass dtypes, 8: no fp16 range scan, 16: dtype fp8 with all 19 eligible convs on e4m3, 32: dtype fp8 with qkv / fc1 / fc2 of the ViT blocks on e4m3)
        self.cfg = cfg
        self.fp8_calibrated = False
        self.fp8_scales = None  # the scales installed last (calibrate_fp8 / set_fp8_calibration): what a rebuilt engine re-installs
        # per-layer precision overrides made through set_layer_precision, in call order: handle state that is NOT part of the
        # packed weights, so every handle that is meant to compute what this one computes (ForwardPipeline.from_engine, an
        # engine rebuilt for a larger input) replays them
        self.layer_precision: Dict[str, int] = {}
        # streams = 0 (the default): the intra-forward schedule is MEASURED at the first forward of at least AUTO_TUNE_MIN_BATCH
        # images (include/dptx.h dptx_tune_schedule; a few extra forwards and one host synchronisation, once per handle)
        self._schedule_tuned = bool(streams) or bool(os.environ.get("DPTX_STREAMS")) or device_id is None
        self.dtype = dtype
        self.backbone = backbone
        self.h = _vp()
        rc = self.lib.dptx_create(C.byref(self.h), C.byref(cfg))
        if rc != 0:
            self.h = None
            raise RuntimeError(f"dptx_create failed: {ERRORS.get(rc, rc)}"
                               + (" (no HIP device visible; the HIP path has no CPU fallback)" if rc == -4 else ""))

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.dptx_last_error(self.h).decode()
            raise RuntimeError(f"{what} failed ({ERRORS.get(rc, rc)}): {msg}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.dptx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights
    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        for k, v in sd.items():
            a = v.detach().to("cpu", torch.float32).contiguous().numpy()
            shape = (C.c_int64 * a.ndim)(*a.shape)
            self._check(self.lib.dptx_load_tensor(self.h, k.encode(), a.ctypes.data, shape, a.ndim), f"load_tensor({k})")
        self._check(self.lib.dptx_finalize_weights(self.h), "finalize_weights")

    @property
    def packed_bytes(self) -> int:
        return self.lib.dptx_packed_bytes(self.h)

    @property
    def workspace_bytes(self) -> int:
        return self.lib.dptx_workspace_bytes(self.h)

    def export_packed_host(self) -> np.ndarray:
        buf = np.empty(self.packed_bytes, dtype=np.uint8)
        self._check(self.lib.dptx_export_packed_host(self.h, buf.ctypes.data, buf.size), "export_packed_host")
        return buf

    def export_packed(self) -> torch.Tensor:
        t = torch.empty(self.packed_bytes, dtype=torch.uint8, device=f"cuda:{self.cfg.device_id}")
        self._check(self.lib.dptx_export_packed_device(self.h, t.data_ptr(), t.numel(), _stream(t.device)), "export_packed_device")
        return t

    def share_weights_from(self, owner: "Engine"):
        """This handle reads `owner`'s packed weights in place (include/dptx.h dptx_share_packed): no second copy on the device.
        The allocation is reference-counted on the C side (handles may be closed in any order); loading / importing weights
        of its own un-shares this handle, and `owner` reloading its weights leaves this handle on the blob it shared."""
        self._check(self.lib.dptx_share_packed(self.h, owner.h), "share_packed")
        self._weights_owner = owner

    def import_packed(self, blob: torch.Tensor):
        assert blob.is_cuda and blob.dtype == torch.uint8 and blob.is_contiguous()
        self._check(self.lib.dptx_import_packed_device(self.h, blob.data_ptr(), blob.numel(), _stream(blob.device)), "import_packed_device")
        torch.cuda.current_stream(blob.device).synchronize()

    # ---- packed-blob cache on disk: skips the fp32 fold / re-layout / upload-from-fp32 at start-up
    def _cache_meta(self) -> Dict[str, str]:
        c = self.cfg
        return {"format": "dptx-packed-v1", "library": self.lib.dptx_version().decode(), "dtype": self.dtype,
                "num_channels": str(c.num_channels), "dual_task": str(c.dual_task), "backbone": self.backbone,
                "ws_form": str(c.ws_form),
                "ws_eps": repr(float(c.ws_eps)), "packed_bytes": str(self.packed_bytes)}

    def save_packed(self, path: str):
        """Writes the packed weight blob (what ranks exchange at start-up) as a safetensors file."""
        from safetensors.torch import save_file
        blob = torch.from_numpy(self.export_packed_host())
        save_file({"blob": blob}, path, metadata=self._cache_meta())

    def load_packed(self, path: str):
        """Loads a blob written by save_packed into this (device) engine; refuses blobs of another configuration."""
        from safetensors import safe_open
        with safe_open(path, framework="pt", device="cpu") as f:
            meta = f.metadata() or {}
            want = self._cache_meta()
            bad = {k: (meta.get(k), v) for k, v in want.items() if meta.get(k) != v}
            if bad:
                raise RuntimeError(f"packed-weight cache {path} does not match this engine: {bad}")
            blob = f.get_tensor("blob")
        self.import_packed(blob.to(f"cuda:{self.cfg.device_id}"))

    # ---- compute
    AUTO_TUNE_MIN_BATCH = 8

    def tune_schedule(self, x: torch.Tensor, out: torch.Tensor, out_depth: Optional[torch.Tensor] = None, reps: int = 2) -> dict:
        """Times this batch's forward under both intra-forward schedules and keeps the faster one (include/dptx.h
        dptx_tune_schedule); blocks until done.  `out` (and `out_depth`) receive a valid result."""
        B, _, H, W = x.shape
        self._check(self.lib.dptx_tune_schedule(self.h, x.data_ptr(), IO_DTYPES[x.dtype], out.data_ptr(), _ptr(out_depth), B, H, W,
                                                int(reps), _stream(x.device)), "tune_schedule")
        self._schedule_tuned = True
        return self.schedule_info()

    def schedule_info(self) -> dict:
        """{'split': two half-batches on two internal streams?, 'tuned': measured?, 'ms_single', 'ms_split'}"""
        sp, tu, a, b = C.c_int32(0), C.c_int32(0), C.c_float(0), C.c_float(0)
        self._check(self.lib.dptx_schedule_info(self.h, C.byref(sp), C.byref(tu), C.byref(a), C.byref(b)), "schedule_info")
        return {"split": bool(sp.value), "tuned": bool(tu.value), "ms_single": round(a.value, 4), "ms_split": round(b.value, 4)}

    def forward(self, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("dptx forward needs a CUDA(HIP) tensor; there is no CPU fallback")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError(f"expected [B,3,H,W] with H, W multiples of 32 (384x384 is the trained size), got {tuple(x.shape)}")
        if x.dtype not in IO_DTYPES:  # fp32 (the reference's convention), bf16 and fp16 images are read as they are
            x = x.float()
        x = x.contiguous()
        B, _, H, W = x.shape
        if x.device.index != self.cfg.device_id:
            raise RuntimeError(f"input is on {x.device}, the engine was created for cuda:{self.cfg.device_id}")
        if out is None:  # the result comes back in the input's element type
            out = torch.empty(B, self.cfg.num_channels, H, W, dtype=x.dtype, device=x.device)
        else:
            _check_out(out, (B, self.cfg.num_channels, H, W), x.device, x.dtype)
        if not self._schedule_tuned and B >= self.AUTO_TUNE_MIN_BATCH:
            self.tune_schedule(x, out)
        self._check(self.lib.dptx_forward_hw(self.h, x.data_ptr(), IO_DTYPES[x.dtype], out.data_ptr(), B, H, W,
                                             _stream(x.device)), "forward")
        return out

    def forward_dual(self, x: torch.Tensor, out_normal: Optional[torch.Tensor] = None,
                     out_depth: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """One encoder pass, two decoders (handle created with dual=True): ([B,3,H,W] normals, [B,1,H,W] depth)."""
        if not x.is_cuda:
            raise RuntimeError("dptx forward needs a CUDA(HIP) tensor; there is no CPU fallback")
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] % 32 or x.shape[3] % 32:
            raise ValueError(f"expected [B,3,H,W] with H, W multiples of 32, got {tuple(x.shape)}")
        if x.dtype not in IO_DTYPES:
            x = x.float()
        x = x.contiguous()
        B, _, H, W = x.shape
        if x.device.index != self.cfg.device_id:
            raise RuntimeError(f"input is on {x.device}, the engine was created for cuda:{self.cfg.device_id}")
        if out_normal is None:
            out_normal = torch.empty(B, 3, H, W, dtype=x.dtype, device=x.device)
        else:
            _check_out(out_normal, (B, 3, H, W), x.device, x.dtype)
        if out_depth is None:
            out_depth = torch.empty(B, 1, H, W, dtype=x.dtype, device=x.device)
        else:
            _check_out(out_depth, (B, 1, H, W), x.device, x.dtype)
        if not self._schedule_tuned and B >= self.AUTO_TUNE_MIN_BATCH:
            self.tune_schedule(x, out_normal, out_depth)
        self._check(self.lib.dptx_forward_dual(self.h, x.data_ptr(), IO_DTYPES[x.dtype], out_normal.data_ptr(), out_depth.data_ptr(), B, H, W,
                                               _stream(x.device)), "forward_dual")
        return out_normal, out_depth

    def set_layer_precision(self, conv_weight_key: str, mfmas: int):
        """dtype 'mixed': run one decoder convolution with 1 or 3 MFMAs per product (include/dptx.h)."""
        self._check(self.lib.dptx_set_layer_precision(self.h, conv_weight_key.encode(), int(mfmas)), "set_layer_precision")
        self.layer_precision.pop(conv_weight_key, None)   # keep call order: a repeated key moves to the end
        self.layer_precision[conv_weight_key] = int(mfmas)

    def copy_layer_precision_from(self, other: "Engine"):
        """Replays `other`'s set_layer_precision calls on this handle (same order)."""
        for k, v in other.layer_precision.items():
            self.set_layer_precision(k, v)

    # ---- fp8 dtype: per-tensor activation scales (include/dptx.h dptx_calibrate_fp8)
    def calibrate_fp8(self, x: torch.Tensor):
        """Measures max |x| of every tensor that gets an e4m3 copy on this batch (one forward with a bf16 decoder) and sets
        the tensors' power-of-two scales.  Returns that forward's result(s)."""
        if not x.is_cuda:
            raise RuntimeError("dptx needs a CUDA(HIP) tensor; there is no CPU fallback")
        if x.dtype not in IO_DTYPES:
            x = x.float()
        x = x.contiguous()
        B, _, H, W = x.shape
        dual = bool(self.cfg.dual_task)
        y = torch.empty(B, 3 if dual else self.cfg.num_channels, H, W, dtype=x.dtype, device=x.device)
        y2 = torch.empty(B, 1, H, W, dtype=x.dtype, device=x.device) if dual else None
        self._check(self.lib.dptx_calibrate_fp8(self.h, x.data_ptr(), IO_DTYPES[x.dtype], y.data_ptr(), _ptr(y2), B, H, W,
                                                _stream(x.device)), "calibrate_fp8")
        self.fp8_calibrated = True
        self.fp8_scales = self.fp8_calibration()[0]
        return (y, y2) if dual else y

    def fp8_calibration(self):
        """(scales, max |x|) of the e4m3 tensors of the last forward, in launch order."""
        s = np.zeros(128, dtype=np.float32)
        a = np.zeros(128, dtype=np.float32)
        n = self.lib.dptx_fp8_get_calibration(self.h, s.ctypes.data_as(_f32p), a.ctypes.data_as(_f32p), 128)
        if n < 0:
            self._check(n, "fp8_get_calibration")
        return s[:n].copy(), a[:n].copy()

    def set_fp8_calibration(self, scales):
        s = np.ascontiguousarray(scales, dtype=np.float32)
        self._check(self.lib.dptx_fp8_set_calibration(self.h, s.ctypes.data_as(_f32p), int(s.size)), "fp8_set_calibration")
        self.fp8_calibrated = True
        self.fp8_scales = s.copy()

    # ---- arena debugging (include/dptx.h dptx_debug_arena_*)
    def arena_fill(self, byte_value: int):
        """Sets every byte of the activation arena to byte_value (0xFF = NaN in every element type) after a device sync."""
        self._check(self.lib.dptx_debug_arena_fill(self.h, int(byte_value)), "debug_arena_fill")

    def arena_layout(self) -> Dict[str, object]:
        n = self.lib.dptx_debug_arena_layout(self.h, None, 0)
        buf = C.create_string_buffer(n)
        self.lib.dptx_debug_arena_layout(self.h, buf, n)
        out: Dict[str, object] = {"bufs": {}}
        for line in buf.value.decode().splitlines():
            f = line.split()
            if f[0] == "buf":
                out["bufs"][f[1]] = (int(f[2]), int(f[3]), int(f[4]))
            else:
                out[f[0]] = int(f[1])
        return out

    def arena_checksums(self) -> torch.Tensor:
        """int64 device tensor [planes, regions, nbuf] of word sums of every arena buffer as the last forward left it (queued
        on the current stream behind that forward; no host sync)."""
        dev = torch.device("cuda", self.cfg.device_id)
        n = self.lib.dptx_debug_arena_checksums(self.h, None, 0, None)
        out = torch.empty(n, dtype=torch.int64, device=dev)
        rc = self.lib.dptx_debug_arena_checksums(self.h, out.data_ptr(), n, _stream(dev))
        if rc < 0:
            self._check(rc, "debug_arena_checksums")
        nbuf = len(self.arena_layout()["bufs"])
        planes = 2 if self.dtype in ("bf16x3", "fp16x3", "mixed", "fp8") else 1
        return out.view(planes, n // (planes * nbuf), nbuf)

    def arena_read(self, offset: int, nbytes: int) -> np.ndarray:
        a = np.empty(nbytes, dtype=np.uint8)
        self._check(self.lib.dptx_debug_arena_read(self.h, a.ctypes.data, offset, nbytes), "debug_arena_read")
        return a

    def range_overflowed(self, reset: bool = True) -> bool:
        """fp16-plane dtypes: has any forward since the last reset left the fp16 range (include/dptx.h dptx_range_status)?
        Waits for the current stream."""
        v = C.c_int32(0)
        self._check(self.lib.dptx_range_status(self.h, C.byref(v), int(reset), _stream(torch.device("cuda", self.cfg.device_id))), "range_status")
        return bool(v.value)

    def enable_taps(self, on: bool = True):
        self._check(self.lib.dptx_enable_taps(self.h, int(on)), "enable_taps")
        self.taps_on = bool(on)

    def tap(self, name: str) -> torch.Tensor:
        """Stage activation of the last forward as an fp32 CPU tensor in NCHW ([B,S,768] for tokens; S = 577 at 384x384)."""
        shape = (C.c_int64 * 4)()
        probe = np.empty(1, dtype=np.float32)
        rc = self.lib.dptx_tap(self.h, name.encode(), probe.ctypes.data, 0, shape)  # size query
        if rc == -2:
            self._check(rc, f"tap({name})")
        b, h, w, c = list(shape)
        buf = np.empty(b * h * w * c, dtype=np.float32)
        self._check(self.lib.dptx_tap(self.h, name.encode(), buf.ctypes.data, buf.size, shape), f"tap({name})")
        t = torch.from_numpy(buf).reshape(b, h, w, c)
        if name.startswith(("tok", "blk")):
            return t.reshape(b, h, w)
        return t.permute(0, 3, 1, 2).contiguous()

    def set_profiling(self, on: bool = True):
        self._check(self.lib.dptx_set_profiling(self.h, int(on)), "set_profiling")

    def profile(self) -> Dict[str, Tuple[float, int, float]]:
        """{category: (ms, launches, executed MACs per image)} of the last profiled forward."""
        out = {}
        for i, name in enumerate(("gemm", "attention", "norm", "other")):
            ms, n, macs = C.c_double(), C.c_int64(), C.c_double()
            self._check(self.lib.dptx_profile_get(self.h, i, C.byref(ms), C.byref(n), C.byref(macs)), "profile_get")
            out[name] = (ms.value, n.value, macs.value)
        return out

    def profile_dump(self, path: str):
        self._check(self.lib.dptx_profile_dump(self.h, path.encode()), "profile_dump")

    def info(self) -> Tuple[int, float, float]:
        n, a, e = C.c_int64(), C.c_double(), C.c_double()
        self._check(self.lib.dptx_forward_info(self.h, C.byref(n), C.byref(a), C.byref(e)), "forward_info")
        return n.value, a.value, e.value
