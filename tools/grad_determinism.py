import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, hashprng as H
import hypernerf_torch_amd as HN
from hypernerf_torch_amd.hypernerf.models import NerfModel
from hypernerf_torch_amd.hypernerf import model_utils
from hypernerf_torch_amd import functional as F
from gpu_common import EMB, load_hash, rays_for
for prec in ("bf16", "fp32"):
    HN.set_precision(prec)
    m = NerfModel(EMB, n_samples_coarse=64, n_samples_fine=64, noise_std=1.0, view_fourier_dim=6, hyper_slice_method="bendy_sheet", use_nerf_embed=True, use_alpha_cond=True)
    load_hash(m, 3); m = m.cuda()
    arena = HN.ParamArena(m.parameters())
    b = 1024
    o, d, idx = rays_for(3, b)
    rays = torch.cat([o, d, torch.zeros(b,1), torch.ones(b,1), idx[:,None].float()], 1).cuda()
    gt = H.uniform(3, "gt", (b,3), 0, 1).cuda()
    rng = {"t_rand": H.uniform(3,"t",(b,64),0,1).cuda(), "u": H.uniform(3,"u",(b,64),0,1).cuda(), "noise_coarse": H.normal(3,"n1",(b,64,1)).cuda(), "noise_fine": H.normal(3,"n2",(b,128,1)).cuda()}
    gs = []
    for it in range(4):
        arena.zero_grad()
        out = m(model_utils.prepare_ray_dict(rays), {}, rng=rng)
        loss = ((out["coarse"]["rgb"]-gt)**2).mean() + ((out["fine"]["rgb"]-gt)**2).mean()
        F.backward(loss)
        torch.cuda.synchronize()
        gs.append(arena.grad.clone())
    names = [(k, p) for k, p in m.named_parameters()]
    diff = [(gs[i] != gs[0]).sum().item() for i in range(1, 4)]
    print(prec, "PARTIALS", os.environ.get("HN_WGRAD_PARTIALS", "1"), "differing elements vs run 0:", diff, "of", gs[0].numel())
    bad = set()
    for k, p in names:
        off = (p.grad.data_ptr() - arena.grad.data_ptr()) // 4
        if (gs[1][off:off+p.numel()] != gs[0][off:off+p.numel()]).any(): bad.add(k)
    print("   tensors that differ:", sorted(bad)[:6], len(bad))
