import sys; sys.path.insert(0, "/root/repo")
import torch, hypernerf_torch_amd as HN
from hypernerf_torch_amd import calibration as CAL
d = torch.device("cuda:0")
for rep in range(3):
    for pat in (0, 1):
        for gib in (4.0, 1.0):
            r = CAL.stream_probe(d, gib=gib, pattern=pat)
            print("pattern", pat, "GiB", gib, "TB/s", round(r["tbps"], 3))
