"""HBM stream probes (csrc/hn_calib.hip) in several forms on one box: what rate the weight-gradient kernel's access
pattern and DMA protocol can reach without its products.   python tools/stream_pattern_probe.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hypernerf_torch_amd as HN  # noqa: F401
from hypernerf_torch_amd import _lib as L, calibration as CAL
d = torch.device("cuda:0")
L.load()
n = 4 << 30
buf = torch.zeros(n, dtype=torch.uint8, device=d)
sink = torch.zeros(4, device=d)


def ring(stages, proto, reps=12):
    t = torch.zeros(16, dtype=torch.int64, device=d)
    for _ in range(reps):
        L.launch("hn_calib_ring", L.ptr(buf), C.c_longlong(n), C.c_int(stages), C.c_int(proto), L.ptr(sink), L.ptr(t), L.stream_handle())
    torch.cuda.synchronize()
    v = t.cpu().tolist()
    return n * v[5] / (v[4] * 1e-8) / 1e12


for rep in range(3):
    print("free-running waves, shared window  ", round(CAL.stream_probe(d, pattern=0)["tbps"], 3), "TB/s")
    print("free-running waves, region per WG   ", round(CAL.stream_probe(d, pattern=1)["tbps"], 3), "TB/s")
    for st in (3, 4, 5):
        print(f"ring of {st} x 32 KiB, barrier per stage", round(ring(st, 0), 3), "TB/s    without the barrier", round(ring(st, 1), 3), "TB/s")
