"""Checkpoint compatibility with the reference (utils/__init__.py:66-88).

Lightning stores the NeRFSystem's weights under `state_dict` with the attribute name of the model as prefix
(`nerf.` for `self.nerf`, train.py:48); the reference strips it with `extract_model_state_dict` and loads the rest
over the model's own state dict.  `NerfModel` here has the same parameter names and shapes, so reference checkpoints
load unchanged — also into a model whose parameters live in a ParamArena (load_state_dict copies in place).
"""
from __future__ import annotations

from typing import Dict, Iterable

import torch


def extract_model_state_dict(ckpt_path, model_name: str = 'model', prefixes_to_ignore: Iterable[str] = ()) -> Dict:
    """{name without '<model_name>.': tensor} for every entry of the checkpoint that belongs to `model_name`.
    Accepts a Lightning checkpoint ({'state_dict': ...}) or a bare state dict (reference: utils/__init__.py:66-81)."""
    blob = torch.load(ckpt_path, map_location=torch.device('cpu'))
    entries = blob['state_dict'] if 'state_dict' in blob else blob
    skip = tuple(prefixes_to_ignore)
    picked = {}
    for key, value in entries.items():
        if not key.startswith(model_name):
            continue
        name = key[len(model_name) + 1:]
        if skip and name.startswith(skip):
            print('ignore', name)
            continue
        picked[name] = value
    return picked


def load_ckpt(model: torch.nn.Module, ckpt_path, model_name: str = 'model', prefixes_to_ignore: Iterable[str] = ()):
    """Load the `model_name.` entries of a checkpoint over the model's current weights; an empty path is a no-op
    (reference: utils/__init__.py:83-89)."""
    if not ckpt_path:
        return
    merged = model.state_dict()
    merged.update(extract_model_state_dict(ckpt_path, model_name, prefixes_to_ignore))
    model.load_state_dict(merged)


def export_lightning_state_dict(model: torch.nn.Module, model_name: str = 'nerf') -> Dict[str, torch.Tensor]:
    """The model's weights under the names Lightning gives them inside the reference's NeRFSystem: every key
    prefixed with the attribute name (`nerf.` for `self.nerf`, train.py:48).  CPU fp32 copies (detached from a
    ParamArena, whose views would otherwise serialise the whole arena per tensor)."""
    return {f"{model_name}.{k}": v.detach().to('cpu', torch.float32).clone() for k, v in model.state_dict().items()}


def save_ckpt(model: torch.nn.Module, ckpt_path, model_name: str = 'nerf', epoch: int = 0, global_step: int = 0,
              optimizer=None, extra: Dict = None):
    """Write a checkpoint the REFERENCE can read back: the layout of the files its ModelCheckpoint callback writes
    (train.py:200-204) as far as its own loader looks at them — {'state_dict': {'<model_name>.<param>': tensor},
    'epoch', 'global_step'} — which `utils.load_ckpt(nerf, path, model_name='nerf')` of the reference
    (utils/__init__.py:66-88, eval.py:135) and `load_ckpt` above both accept.  `optimizer` (an ArenaAdam) is stored
    under 'hn_optimizer' for resuming here; the reference ignores unknown keys."""
    blob = {'state_dict': export_lightning_state_dict(model, model_name), 'epoch': int(epoch),
            'global_step': int(global_step)}
    if optimizer is not None:
        sd = optimizer.state_dict()
        blob['hn_optimizer'] = {k: (v.detach().cpu().clone() if isinstance(v, torch.Tensor) else v)
                                for k, v in sd.items()}
    if extra:
        blob.update(extra)
    torch.save(blob, ckpt_path)
    return ckpt_path
