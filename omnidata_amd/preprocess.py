"""Pre/post-processing of demo.py restated without torchvision (absent in this image).

Follows omnidata_tools/torch/demo.py:74-76,92-95 (Resize(384,BILINEAR) on the shorter side ->
CenterCrop(384) -> ToTensor [-> Normalize(0.5,0.5) for depth]), :101-102 (512 RGB preview),
:137-150 (grey -> 3 channels, clamp, ToPILImage / bicubic + 1-x + viridis).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image


def resize_shorter(img: Image.Image, size: int) -> Image.Image:
    w, h = img.size
    if (w <= h and w == size) or (h <= w and h == size):
        return img
    if w < h:
        ow, oh = size, int(size * h / w)
    else:
        oh, ow = size, int(size * w / h)
    return img.resize((ow, oh), Image.BILINEAR)


def center_crop(img: Image.Image, size: int) -> Image.Image:
    w, h = img.size
    if w < size or h < size:  # torchvision pads with 0 first
        pl, pt = max((size - w) // 2, 0), max((size - h) // 2, 0)
        canvas = Image.new(img.mode, (max(w, size), max(h, size)))
        canvas.paste(img, (pl, pt))
        img = canvas
        w, h = img.size
    top, left = int(round((h - size) / 2.0)), int(round((w - size) / 2.0))
    return img.crop((left, top, left + size, top + size))


def to_tensor(img: Image.Image) -> torch.Tensor:
    a = np.array(img)  # writable copy
    if a.ndim == 2:
        a = a[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1)
    if t.dtype == torch.uint8:
        return t.float().div(255.0)
    return t.float()


def image_to_input(img: Image.Image, task: str, image_size: int = 384) -> torch.Tensor:
    """-> [1,3,384,384] fp32 in the model's input convention."""
    t = to_tensor(center_crop(resize_shorter(img, image_size), image_size))
    if task == "depth":
        t = (t - 0.5) / 0.5
    t = t[:3].unsqueeze(0)
    if t.shape[1] == 1:
        t = t.repeat_interleave(3, 1)
    return t


def rgb_preview(img: Image.Image) -> Image.Image:
    return center_crop(resize_shorter(img, 512), 512)


def normal_to_pil(output: torch.Tensor) -> Image.Image:
    """ToPILImage of a clamped [3,H,W] float tensor: mul(255).byte() (truncation)."""
    a = output.detach().cpu().clamp(0, 1).mul(255).byte().permute(1, 2, 0).numpy()
    return Image.fromarray(a)


def colorize_viridis(o: np.ndarray) -> np.ndarray:
    """plt.imsave(cmap='viridis') of a [H,W] array: normalise to the data range, apply the colormap, RGBA uint8."""
    from matplotlib import cm
    lo, hi = float(o.min()), float(o.max())
    n = (o - lo) / (hi - lo) if hi > lo else np.zeros_like(o)
    return (cm.get_cmap("viridis")(n) * 255).astype(np.uint8)


def depth_to_rgba(output: torch.Tensor) -> np.ndarray:
    """[1,H,W] or [H,W] clamped depth -> bicubic 512x512 -> clamp -> 1-x -> viridis RGBA uint8."""
    from matplotlib import cm
    o = output.detach().float().cpu().reshape(1, 1, *output.shape[-2:])
    o = F.interpolate(o, (512, 512), mode="bicubic").clamp(0, 1)
    o = (1 - o).squeeze().numpy()
    lo, hi = float(o.min()), float(o.max())  # plt.imsave normalises to [vmin, vmax] = data range
    n = (o - lo) / (hi - lo) if hi > lo else np.zeros_like(o)
    return (cm.get_cmap("viridis")(n) * 255).astype(np.uint8)


# ------------------------------------------------------------------ GPU path (libdptx.so prepost kernels)
def image_to_input_gpu(img, task: str, device="cuda:0") -> torch.Tensor:
    """Same result as image_to_input() (bit-identical), computed on the GPU from the raw uint8 pixels: only the
    undecoded image crosses PCIe.  RGB / greyscale uint8 images; other modes (RGBA is premultiplied by Pillow's
    resize) take the PIL path."""
    from .engine import load_library
    if isinstance(img, Image.Image):
        if img.mode not in ("RGB", "L"):
            return image_to_input(img, task).to(device)
        a = torch.from_numpy(np.array(img))
    else:
        a = img
    if a.dim() == 2:
        a = a[:, :, None]
    assert a.dtype == torch.uint8 and a.shape[2] in (1, 3)
    a = a.to(device).contiguous()
    H, W, C = a.shape
    x = torch.empty(1, 3, 384, 384, dtype=torch.float32, device=device)
    rc = load_library().dptx_preprocess_u8(a.data_ptr(), H, W, C, W * C, int(task == "depth"), x.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError(f"dptx_preprocess_u8 failed ({rc})")
    return x


def normal_to_u8_gpu(output: torch.Tensor) -> torch.Tensor:
    """[3,384,384] float (cuda) -> [384,384,3] uint8 (cuda): clamp(0,1)*255 truncated, as ToPILImage does."""
    from .engine import load_library
    y = output.detach().float().contiguous()
    out = torch.empty(384, 384, 3, dtype=torch.uint8, device=y.device)
    rc = load_library().dptx_postprocess_normal_u8(y.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError(f"dptx_postprocess_normal_u8 failed ({rc})")
    return out


def depth_to_512_gpu(output: torch.Tensor) -> torch.Tensor:
    """[384,384] float (cuda) -> [512,512] float (cuda): bicubic, clamp(0,1), 1-x (demo.py:143-145)."""
    from .engine import load_library
    y = output.detach().float().reshape(384, 384).contiguous()
    out = torch.empty(512, 512, dtype=torch.float32, device=y.device)
    rc = load_library().dptx_postprocess_depth(y.data_ptr(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    if rc != 0:
        raise RuntimeError(f"dptx_postprocess_depth failed ({rc})")
    return out
