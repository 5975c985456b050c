"""Box calibration for benchmark lines (csrc/hn_calib.hip): what THIS GPU sustains, measured in the same process as the
timed region — a register-resident MFMA probe (dense bf16 TFLOP/s + the shader clock it holds), an LDS-DMA stream probe
(HBM TB/s of the weight-gradient kernel's access pattern) and the board's hwmon power / clock while the timed steps run.
The pool's boxes differ by several per cent and a training step runs against the power cap; a line that carries these
can be compared with a line from another box (value / probe), one that does not cannot.  Measurement infrastructure:
nothing here is on the render path."""
from __future__ import annotations

import ctypes as C
import glob
import os
import threading
import time
from typing import Dict, List, Optional

import torch

from . import _lib as L

MFMA_FLOPS_PER_ITER = 512 * 8 * 8 * 32768.0        # workgroups x waves x accumulators x FLOPs of one 32x32x16 MFMA


def _ticks(t: torch.Tensor) -> List[int]:
    torch.cuda.synchronize()
    return t.cpu().tolist()


def mfma_probe(device, target_ms: float = 50.0) -> Dict[str, float]:
    """Sustained dense bf16 MFMA rate and the shader clock held while doing it (hn_calib_mfma)."""
    L.load()
    sink = torch.zeros(4, dtype=torch.float32, device=device)

    def run(iters: int):
        t = torch.zeros(16, dtype=torch.int64, device=device)
        L.launch("hn_calib_mfma", C.c_int(iters), L.ptr(sink), L.ptr(t), L.stream_handle())
        v = _ticks(t)
        return v[4] * L.TIMELINE_TICK_S, v[8], v[9]
    run(2000)                                        # wake the clocks
    s, _, _ = run(4000)
    iters = max(4000, min(400000, int(4000 * target_ms * 1e-3 / max(s, 1e-6))))
    s, sclk_ticks, wall_ticks = run(iters)
    return {"tflops": MFMA_FLOPS_PER_ITER * iters / s / 1e12, "ms": s * 1e3, "iters": iters,
            "sclk_mhz": 100.0 * sclk_ticks / max(1, wall_ticks)}


def stream_probe(device, gib: float = 4.0, target_ms: float = 40.0, pattern: int = 0) -> Dict[str, float]:
    """HBM rate of an LDS-DMA stream over a buffer far larger than the 256 MB Infinity Cache (hn_calib_stream).
    pattern 0: the workgroups share one moving window; 1: a contiguous region per workgroup (the weight-gradient jobs)."""
    L.load()
    n = int(gib * (1 << 30)) // 65536 * 65536
    buf = torch.empty(n, dtype=torch.uint8, device=device)
    buf.zero_()
    sink = torch.zeros(4, dtype=torch.float32, device=device)
    t = torch.zeros(16, dtype=torch.int64, device=device)
    def go():
        L.launch("hn_calib_stream_pattern", L.ptr(buf), C.c_longlong(n), C.c_int(pattern), L.ptr(sink), L.ptr(t),
                 L.stream_handle())
    go()
    first = _ticks(t)[4] * L.TIMELINE_TICK_S
    reps = max(2, min(200, int(target_ms * 1e-3 / max(first, 1e-6))))
    t.zero_()
    for _ in range(reps):
        go()
    v = _ticks(t)
    s = v[4] * L.TIMELINE_TICK_S
    del buf
    return {"tbps": n * float(v[5]) / s / 1e12, "ms": s * 1e3, "passes": int(v[5]), "gib_per_pass": n / float(1 << 30)}


# ---- hwmon --------------------------------------------------------------------------------------
def _rd(path: str) -> Optional[int]:
    try:
        with open(path) as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return None


def hwmon_dirs() -> List[str]:
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        if any(os.path.exists(f"{d}/{f}") for f in ("power1_average", "power1_input")):
            out.append(d)
    return out


def _board_of_device(device) -> Optional[str]:
    """hwmon directory of the torch device, by PCI address (None if the runtime does not say)."""
    try:
        pr = torch.cuda.get_device_properties(device)
        want = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}"
    except Exception:       # noqa: BLE001 (older property sets)
        return None
    for d in hwmon_dirs():
        real = os.path.realpath(os.path.join(d, "..", ".."))
        if want in real.lower():
            return d
    return None


class PowerSampler:
    """Samples every amdgpu hwmon (power, sclk, mclk) from a host thread while a timed region runs; `summary()` reports
    the board of `device` (by PCI address, else the board whose power rose most over its first sample — a shared node
    exposes all eight boards)."""

    def __init__(self, device, period_s: float = 0.02):
        self.dirs = hwmon_dirs()
        self.mine = _board_of_device(device)
        self.period = period_s
        self.rows: List[Dict[str, int]] = []
        self._stop = threading.Event()
        self._th: Optional[threading.Thread] = None

    def _sample(self):
        row = {}
        for i, d in enumerate(self.dirs):
            p = _rd(f"{d}/power1_average")
            if p is None:
                p = _rd(f"{d}/power1_input")
            row[f"p{i}"] = p or 0
            row[f"s{i}"] = _rd(f"{d}/freq1_input") or 0
            row[f"m{i}"] = _rd(f"{d}/freq2_input") or 0
        self.rows.append(row)

    def __enter__(self):
        if self.dirs:
            self._sample()
            self._th = threading.Thread(target=self._loop, daemon=True)
            self._th.start()
        return self

    def _loop(self):
        while not self._stop.wait(self.period):
            self._sample()

    def __exit__(self, *exc):
        self._stop.set()
        if self._th is not None:
            self._th.join(timeout=1.0)
        return False

    def summary(self) -> Dict[str, object]:
        if not self.dirs or len(self.rows) < 2:
            return {"power_w": None, "sclk_mhz": None, "note": "no amdgpu hwmon files readable on this box"}
        if self.mine in self.dirs:
            i, how = self.dirs.index(self.mine), "PCI address of the device"
        else:
            rise = [max(r[f"p{k}"] for r in self.rows[1:]) - self.rows[0][f"p{k}"] for k in range(len(self.dirs))]
            i, how = max(range(len(rise)), key=rise.__getitem__), "board whose power rose most during the region"
        busy = self.rows[1:]
        pw = [r[f"p{i}"] / 1e6 for r in busy]
        sc = [r[f"s{i}"] / 1e6 for r in busy if r[f"s{i}"]]
        cap = _rd(f"{self.dirs[i]}/power1_cap")
        return {"power_w": sum(pw) / len(pw), "power_w_max": max(pw), "power_cap_w": cap / 1e6 if cap else None,
                "sclk_mhz": (sum(sc) / len(sc)) if sc else None, "sclk_mhz_min_max": [min(sc), max(sc)] if sc else None,
                "mclk_mhz": sorted({round(r[f"m{i}"] / 1e6) for r in busy if r[f"m{i}"]}), "samples": len(busy),
                "board": os.path.basename(os.path.dirname(os.path.dirname(os.path.dirname(self.dirs[i])))), "board_by": how,
                "boards_visible": len(self.dirs)}


def probes(device) -> Dict[str, object]:
    """{mfma_probe_tflops, mfma_probe_sclk_mhz, hbm_probe_tbps, ...}: ~0.1 s of GPU time."""
    m = mfma_probe(device)
    s = stream_probe(device)
    return {"mfma_probe_tflops": m["tflops"], "mfma_probe_frac_of_2.5PF": m["tflops"] / 2500.0,
            "mfma_probe_sclk_mhz": m["sclk_mhz"], "mfma_probe_ms": m["ms"],
            "hbm_probe_tbps": s["tbps"], "hbm_probe_frac_of_8TBps": s["tbps"] / 8.0, "hbm_probe_ms": s["ms"],
            "hbm_probe_gib_per_pass": s["gib_per_pass"],
            "what": "same process, right before the timed region: hn_calib_mfma (register-resident "
                    "v_mfma_f32_32x32x16_bf16, 4 waves per SIMD; sclk = s_memtime ticks per 100 MHz wall tick) and "
                    "hn_calib_stream (LDS-DMA stream of a 4 GiB buffer, the weight-gradient kernel's access pattern)"}
