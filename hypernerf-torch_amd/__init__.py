"""hypernerf_torch_amd — MI355X-native (gfx950) render hot path of songrise/HyperNeRF-torch.

Drop-in module layout (same import paths as the reference, under this package):
    hypernerf_torch_amd.hypernerf.models.NerfModel          <- hypernerf/models.py
    hypernerf_torch_amd.hypernerf.modules.{MLP,GLOEmbed,NerfMLP,HyperSheetMLP}
    hypernerf_torch_amd.hypernerf.warping.{TranslationField,SE3Field}
    hypernerf_torch_amd.hypernerf.model_utils.*
    hypernerf_torch_amd.models.rendering.render_rays        <- models/rendering.py (nerf_pl signature)
    hypernerf_torch_amd.models.nerf.{Embedding,NeRF}        <- models/nerf.py

All device arithmetic runs in hand-written HIP kernels behind the C ABI in include/hn_kernels.h
(csrc/libhn_hip.so).  There is no CPU execution path: ops raise on CPU tensors or a missing library.
"""
from . import _lib
from .arena import ParamArena
from .optim import ArenaAdam
from .functional import get_precision, set_precision

__version__ = "0.1.0"


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP extension for gfx950 (hipcc cross-compiles without a GPU)."""
    return _lib.build(force=force, verbose=verbose)
