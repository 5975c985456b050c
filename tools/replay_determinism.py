"""Diagnostic: 2,000 replays of a captured TrainStep with lr = 0 on identical inputs — the logged loss must be
bit-identical in every replay (the forward has no atomics), and a model used from two streams must agree with itself."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import hypernerf_torch_amd as HN                                  # noqa: E402
from hypernerf_torch_amd.hypernerf.models import NerfModel        # noqa: E402
from hypernerf_torch_amd.hypernerf import model_utils             # noqa: E402
from hypernerf_torch_amd.training import TrainStep                # noqa: E402
from gpu_common import EMB                                        # noqa: E402
import psnr_parity                                                # noqa: E402

dev = "cuda:0"
HN.set_precision("bf16")
torch.manual_seed(0)
kw = dict(near=0.2, far=2.0, n_samples_coarse=64, n_samples_fine=64, noise_std=None, hyper_slice_method="bendy_sheet",
          use_warp=True, use_nerf_embed=True, use_alpha_cond=True, view_fourier_dim=6)
m = NerfModel(EMB, **kw).to(dev)
m.use_stratified_sampling = False
g = torch.Generator().manual_seed(7)
rays, col = (x.to(dev) for x in psnr_parity.scene(1024, g))
ts = TrainStep(m, lr=0.0)
first = None
bad = 0
for it in range(2000):
    loss = ts.step(rays, col)["train/loss"]
    if first is None:
        first = loss.clone()
    elif not torch.equal(loss, first):
        bad += 1
print("replays 2000, loss deviations:", bad, "loss", float(first))

# two streams: a fresh model whose FIRST forward (table uploads, packing) runs on stream A and whose second runs on
# stream B right away
torch.manual_seed(0)
m2 = NerfModel(EMB, **kw).to(dev)
m2.use_stratified_sampling = False
m2.load_state_dict(m.state_dict())
rd = model_utils.prepare_ray_dict(rays)
extra = {'nerf_alpha': None, 'warp_alpha': None, 'hyper_alpha': None, 'hyper_sheet_alpha': None}
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.synchronize()
with torch.no_grad():
    with torch.cuda.stream(sa):
        a = m2(rd, extra)["fine"]["rgb"]
    with torch.cuda.stream(sb):
        b = m2(rd, extra)["fine"]["rgb"]
    torch.cuda.synchronize()
    c = m(rd, extra)["fine"]["rgb"]
    torch.cuda.synchronize()
print("two streams agree:", torch.equal(a, b), "and with the other model:", torch.equal(a, c))
