"""Shader clock / board power / temperature of one GPU while a loop runs, read from the amdgpu driver's sysfs files by a
sampling thread (no GPU work of its own: nothing is launched, so the measured loop is not perturbed).

Why bench.py carries this: the chip is power-managed -- the same kernels run at 1.8-2.1 GHz depending on
what the loop draws and on the box -- so a throughput number without the clock it was measured at cannot be compared with
another box's, or with last round's.  Every file is optional: a box that hides them yields `available: False`, never an
error.  Files (per card, /sys/class/drm/card*/device): hwmon/hwmon*/freq1_input (sclk, Hz), power1_average | power1_input
(microwatts), power1_cap, temp*_input (millidegrees; labels edge / junction / mem), gpu_busy_percent; pp_dpm_sclk as the
fallback for the clock (the level marked '*').
"""
from __future__ import annotations

import glob
import os
import re
import threading
import time
from typing import Dict, List, Optional


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def _read_num(path: str) -> Optional[float]:
    s = _read(path)
    try:
        return float(s) if s is not None else None
    except ValueError:
        return None


def find_card(pci_bus_id: Optional[str] = None, root: str = "/sys/class/drm") -> Optional[str]:
    """/sys/class/drm/cardN/device of the amdgpu card with this PCI address ('0000:c1:00.0'; case-insensitive), or of the
    only amdgpu card when no address is given / matched and there is exactly one."""
    cards = []
    for dev in sorted(glob.glob(os.path.join(root, "card[0-9]*", "device"))):
        if re.search(r"card\d+-", dev):
            continue   # connectors (card0-DP-1)
        drv = os.path.basename(os.path.realpath(os.path.join(dev, "driver")))
        if drv and drv != "amdgpu":
            continue
        cards.append(dev)
    if pci_bus_id:
        want = pci_bus_id.lower()
        for dev in cards:
            if os.path.basename(os.path.realpath(dev)).lower() == want:
                return dev
    return cards[0] if len(cards) == 1 else None


def device_pci_bus_id(device_index: int) -> Optional[str]:
    """PCI address of a HIP device as sysfs spells it, from torch's device properties (None when torch does not expose it)."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        return f"{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}.0"
    except Exception:
        return None


class GpuTelemetry:
    """with GpuTelemetry(0) as t: ...loop...  ->  t.summary()"""

    def __init__(self, device_index: int = 0, interval_s: float = 0.02, card_dir: Optional[str] = None):
        self.interval = float(interval_s)
        self.card = card_dir if card_dir is not None else find_card(device_pci_bus_id(device_index))
        self.files: Dict[str, str] = {}
        self.samples: Dict[str, List[float]] = {}
        self._stop = threading.Event()
        self._thread: Optional[threading.Thread] = None
        self.t0 = self.t1 = 0.0
        if self.card:
            hw = sorted(glob.glob(os.path.join(self.card, "hwmon", "hwmon*")))
            if hw:
                h = hw[0]
                for key, names in (("sclk_hz", ("freq1_input",)), ("power_uw", ("power1_average", "power1_input")),
                                   ("power_cap_uw", ("power1_cap",))):
                    for n in names:
                        if _read_num(os.path.join(h, n)) is not None:
                            self.files[key] = os.path.join(h, n)
                            break
                for tfile in sorted(glob.glob(os.path.join(h, "temp*_input"))):
                    label = _read(tfile.replace("_input", "_label")) or os.path.basename(tfile)[:5]
                    if _read_num(tfile) is not None:
                        self.files[f"temp_{label}_mc"] = tfile
            if _read_num(os.path.join(self.card, "gpu_busy_percent")) is not None:
                self.files["busy_pct"] = os.path.join(self.card, "gpu_busy_percent")
            if "sclk_hz" not in self.files and _read(os.path.join(self.card, "pp_dpm_sclk")):
                self.files["dpm_sclk"] = os.path.join(self.card, "pp_dpm_sclk")

    @property
    def available(self) -> bool:
        return any(k in self.files for k in ("sclk_hz", "dpm_sclk", "power_uw"))

    def _sample(self):
        for key, path in self.files.items():
            if key == "power_cap_uw":
                continue
            if key == "dpm_sclk":
                txt = _read(path) or ""
                m = re.search(r"(\d+)\s*Mhz\s*\*", txt, re.I)
                if m:
                    self.samples.setdefault("sclk_hz", []).append(float(m.group(1)) * 1e6)
                continue
            v = _read_num(path)
            if v is not None:
                self.samples.setdefault(key, []).append(v)

    def _run(self):
        while not self._stop.is_set():
            self._sample()
            self._stop.wait(self.interval)

    def start(self):
        self.samples = {}
        self._stop.clear()
        self.t0 = time.perf_counter()
        if self.available:
            self._thread = threading.Thread(target=self._run, name="gpu-telemetry", daemon=True)
            self._thread.start()
        return self

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join(timeout=1.0)
            self._thread = None
        self.t1 = time.perf_counter()
        return self

    __enter__ = start

    def __exit__(self, *exc):
        self.stop()

    def read_once(self) -> Dict[str, float]:
        """One reading outside a loop (the idle state before it)."""
        self.samples = {}
        self._sample()
        return self.summary(single=True)

    def summary(self, single: bool = False) -> Dict[str, object]:
        if not self.available:
            return {"available": False, "note": "no amdgpu sysfs telemetry readable for this device (card: %s)" % self.card}

        def stats(vals, scale):
            if not vals:
                return None
            if single:
                return round(vals[0] * scale, 1)
            return {"mean": round(sum(vals) / len(vals) * scale, 1), "min": round(min(vals) * scale, 1), "max": round(max(vals) * scale, 1)}
        out: Dict[str, object] = {"available": True, "source": "amdgpu sysfs (" + self.card + ")"}
        out["sclk_mhz"] = stats(self.samples.get("sclk_hz", []), 1e-6)
        out["power_w"] = stats(self.samples.get("power_uw", []), 1e-6)
        for k, v in self.samples.items():
            if k.startswith("temp_"):
                out[k.replace("_mc", "_c")] = stats(v, 1e-3)
        if "busy_pct" in self.samples:
            out["busy_pct"] = stats(self.samples["busy_pct"], 1.0)
        cap = _read_num(self.files["power_cap_uw"]) if "power_cap_uw" in self.files else None
        if cap:
            out["power_cap_w"] = round(cap * 1e-6, 1)
        if not single:
            out["samples"] = max((len(v) for v in self.samples.values()), default=0)
            out["window_s"] = round(self.t1 - self.t0, 3)
            out["interval_s"] = self.interval
        return out
