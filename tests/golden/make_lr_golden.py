#!/usr/bin/env python3
"""Golden learning-rate sequences from the REFERENCE's own `utils.get_scheduler` (utils/__init__.py:43-59, with its
GradualWarmupScheduler, utils/warmup_scheduler.py), build container only:

    python tests/golden/make_lr_golden.py        -> tests/golden/g17_lr_schedules.npz

For each hyper-parameter set the reference's scheduler is built on a torch Adam (lr 5e-4, the reference's default) and
stepped once per epoch, as Lightning does with what `configure_optimizers` returns (train.py:116-131); the fixture
holds the learning rate the optimizer has DURING each epoch (index 0 = before the first scheduler.step()).
'poly' is not recorded: the reference raises NameError there (`LambdaLR` is never imported, utils/__init__.py:51).
Import shims as in check_ckpt_with_reference.py (the reference's utils package pulls in its visualisation helpers)."""
import json
import os
import sys
import types
import warnings

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("HN_REFERENCE", "/root/reference")

import numpy as np
import torch


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


tv = _stub("torchvision")
tv.transforms = _stub("torchvision.transforms")
_stub("cv2", COLORMAP_JET=2)
pil = _stub("PIL")
pil.Image = _stub("PIL.Image")
sys.path.insert(0, REF)
import utils as R_utils            # noqa: E402

CASES = {
    "steplr": dict(lr_scheduler="steplr", decay_step=[3, 7], decay_gamma=0.1, warmup_epochs=0),
    "steplr_g5": dict(lr_scheduler="steplr", decay_step=[2, 4, 9], decay_gamma=0.5, warmup_epochs=0),
    "cosine": dict(lr_scheduler="cosine", num_epochs=12, warmup_epochs=0),
    "warm_steplr": dict(lr_scheduler="steplr", decay_step=[3, 7], decay_gamma=0.1, warmup_epochs=3, warmup_multiplier=2.0),
    "warm_cosine": dict(lr_scheduler="cosine", num_epochs=12, warmup_epochs=2, warmup_multiplier=1.0),
    "warm_cosine_m4": dict(lr_scheduler="cosine", num_epochs=12, warmup_epochs=4, warmup_multiplier=4.0),
}
N_EPOCHS = 12


def main():
    out = {}
    for name, hp in CASES.items():
        h = types.SimpleNamespace(optimizer="adam", lr=5e-4, momentum=0.9, weight_decay=0.0, **hp)
        p = torch.nn.Parameter(torch.zeros(3))
        opt = R_utils.get_optimizer(h, torch.nn.ModuleList([torch.nn.Linear(2, 2)]))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sch = R_utils.get_scheduler(h, opt)
            lrs = [R_utils.get_learning_rate(opt)]
            for _ in range(N_EPOCHS):
                opt.step()
                sch.step()
                lrs.append(R_utils.get_learning_rate(opt))
        out[name] = np.asarray(lrs, dtype=np.float64)
        print(name, np.array2string(out[name], precision=6))
    out["cases"] = np.asarray(json.dumps(CASES))
    np.savez(os.path.join(HERE, "g17_lr_schedules.npz"), **out)


if __name__ == "__main__":
    main()
