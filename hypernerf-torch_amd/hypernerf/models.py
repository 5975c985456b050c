"""MI355X drop-in for the reference's `hypernerf/models.py`: `NerfModel` and `filter_sigma`.

Constructor arguments, attribute / sub-module names (=> `state_dict` keys), forward signature and the
returned dict follow hypernerf/models.py:111-127, 673-780.  The render itself is a short sequence of
fused HIP launches per level: sampling -> warp-field machine -> hyper-sheet machine -> template machine
(encoders fused as generated features) -> compositing kernel; inverse-CDF sampling between the levels.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
from typing import Any, Callable, Dict, Mapping, Optional, Sequence

import torch
import torch.nn as nn

from .. import _lib as L
from .. import functional as F
from ..machine import AuxSpec, GradIn, OutSpec, Program, copy_features, posenc_features
from . import model_utils, modules, warping


def filter_sigma(points, sigma, render_opts):
    """Dust-threshold / bounding-box masking of densities (reference: hypernerf/models.py:35-63).
    `render_opts` is None on every call the reference makes; the masks are plain tensor ops."""
    if render_opts is None:
        return sigma
    if 'dust_threshold' in render_opts:
        sigma = (sigma >= render_opts.get('dust_threshold', 0.0)) * sigma
    if 'bounding_box' in render_opts:
        xmin, xmax, ymin, ymax, zmin, zmax = render_opts['bounding_box']
        m = ((points[..., 0] >= xmin) & (points[..., 0] <= xmax) & (points[..., 1] >= ymin)
             & (points[..., 1] <= ymax) & (points[..., 2] >= zmin) & (points[..., 2] <= zmax))
        sigma = m * sigma
    return sigma


class NerfModel(nn.Module):
    """HyperNeRF model with coarse and fine template MLPs (reference: hypernerf/models.py:67-780)."""

    def __init__(self, embeddings_dict, near: float = 0.0, far: float = 1.0, n_samples_coarse: int = 64,
                 n_samples_fine: int = 128, noise_std: float = None, use_warp: bool = True,
                 use_nerf_embed: bool = True, use_alpha_cond: bool = True, use_rgb_cond: bool = False,
                 hyper_slice_method: str = None, hyper_slice_out_dim: int = 4, GLO_dim: int = 8,
                 share_GLO: bool = True, xyz_fourier_dim: int = 10, hyper_fourier_dim: int = 6,
                 view_fourier_dim: int = 4):
        super().__init__()
        self.embeddings_dict: Mapping[str, Sequence[int]] = embeddings_dict
        self.near, self.far = near, far
        self.use_viewdirs: bool = True
        self.noise_std = noise_std
        self.nerf_trunk_depth: int = 8
        self.nerf_trunk_width: int = 256
        self.nerf_rgb_branch_depth: int = 4
        self.nerf_rgb_branch_width: int = 128
        self.nerf_skips = [4, ]
        self.num_coarse_samples = n_samples_coarse
        self.num_fine_samples = n_samples_fine
        self.use_stratified_sampling: bool = True
        self.use_white_background: bool = False
        self.use_linear_disparity: bool = False
        self.use_sample_at_infinity: bool = True
        self.spatial_point_min_deg, self.spatial_point_max_deg = 0, 10
        self.hyper_point_min_deg, self.hyper_point_max_deg = 0, 4
        self.viewdir_min_deg, self.viewdir_max_deg = 0, 4
        self.use_posenc_identity: bool = True
        self.alpha_channels: int = 1
        self.rgb_channels: int = 3
        self.activation = nn.ReLU()
        self.sigma_activation = nn.Softplus()
        self.rgb_activation = nn.Sigmoid()
        if not share_GLO:
            # the reference leaves nerf_use_warp_embed / hyper_use_warp_embed unbound (models.py:167-174,186)
            raise UnboundLocalError("share_GLO=False is not constructible in the reference (models.py:167-186): "
                                    "local variable 'nerf_use_warp_embed' referenced before assignment")
        nerf_use_warp_embed = hyper_use_warp_embed = use_warp
        self.use_nerf_embed: bool = use_nerf_embed
        self.nerf_embed_cls: Callable[..., nn.Module] = functools.partial(modules.GLOEmbed, embedding_dim=GLO_dim)
        self.nerf_embed_key: str = 'warp'
        self.nerf_use_warp_embed: bool = nerf_use_warp_embed
        self.use_alpha_condition: bool = use_alpha_cond
        self.use_rgb_condition: bool = use_rgb_cond
        self.hyper_slice_method = 'none' if hyper_slice_method is None else hyper_slice_method
        self.hyper_embed_cls: Callable[..., nn.Module] = functools.partial(modules.GLOEmbed, embedding_dim=GLO_dim)
        self.hyper_embed_key: str = 'time'
        self.hyper_use_warp_embed: bool = hyper_use_warp_embed
        self.hyper_sheet_mlp_cls: Callable[..., nn.Module] = modules.HyperSheetMLP
        self.hyper_sheet_out_dim: int = hyper_slice_out_dim
        self.use_warp: bool = use_warp
        self.warp_field_cls: Callable[..., nn.Module] = warping.TranslationField
        self.warp_embed_cls: Callable[..., nn.Module] = functools.partial(modules.GLOEmbed, embedding_dim=GLO_dim)
        self.warp_embed_key: str = 'time'
        self.xyz_freq, self.dir_freq, self.hyper_freq = xyz_fourier_dim, view_fourier_dim, hyper_fourier_dim
        self.GLO_dim = GLO_dim

        if self.use_nerf_embed and not (self.use_rgb_condition or self.use_alpha_condition):
            raise ValueError('Template metadata is enabled but none of the condition'
                             'branches are.')
        if self.use_nerf_embed:
            self.nerf_embed = self.nerf_embed_cls(num_embeddings=max(self.embeddings_dict[self.nerf_embed_key]) + 1)
        if self.use_warp:
            self.warp_embed = self.warp_embed_cls(num_embeddings=max(self.embeddings_dict[self.warp_embed_key]) + 1)
        if self.hyper_slice_method == 'axis_aligned_plane':
            self.hyper_embed = self.hyper_embed_cls(
                num_embeddings=max(self.embeddings_dict[self.hyper_embed_key]) + 1)
        elif self.hyper_slice_method == 'bendy_sheet':
            if not self.hyper_use_warp_embed:
                self.hyper_embed = self.hyper_embed_cls(
                    num_embeddings=max(self.embeddings_dict[self.hyper_embed_key]) + 1)
            self.hyper_sheet_mlp = self.hyper_sheet_mlp_cls(out_ch=self.hyper_sheet_out_dim, in_ch_embed=GLO_dim)
        if self.use_warp:
            self.warp_field = warping.TranslationField(in_ch=3, in_ch_embed=GLO_dim)
        self.alpha_default = 0.0

        self.nerf_in_ch_pos = model_utils.get_posenc_ch_orig(3, self.xyz_freq)
        self.nerf_cond_ch_rgb = model_utils.get_posenc_ch_orig(3, self.dir_freq)
        self.hyper_feat_ch = model_utils.get_posenc_ch_orig(self.hyper_sheet_out_dim, self.hyper_freq)
        if self.use_warp:
            self.nerf_in_ch_pos += self.hyper_feat_ch
        if self.use_rgb_condition:
            self.nerf_cond_ch_rgb += GLO_dim

        def make_mlp():
            return modules.NerfMLP(in_ch=self.nerf_in_ch_pos, trunk_depth=self.nerf_trunk_depth,
                                   trunk_width=self.nerf_trunk_width, rgb_branch_depth=self.nerf_rgb_branch_depth,
                                   rgb_branch_width=self.nerf_rgb_branch_width, hidden_activation=self.activation,
                                   norm=None, skips=self.nerf_skips, alpha_channels=self.alpha_channels,
                                   rgb_channels=self.rgb_channels, rgb_activation=self.rgb_activation,
                                   alpha_condition_dim=GLO_dim if self.use_nerf_embed else 0,
                                   rgb_condition_dim=self.nerf_cond_ch_rgb)

        nerf_mlps_coarse = make_mlp()
        if self.num_fine_samples > 0:
            nerf_mlps_fine = make_mlp()
        else:
            raise UnboundLocalError("n_samples_fine=0 is not constructible in the reference (models.py:292-309): "
                                    "local variable 'nerf_mlps_fine' referenced before assignment; use the legacy "
                                    "render_rays(N_importance=0) path for coarse-only rendering")
        self.nerf_mlps_coarse = nerf_mlps_coarse
        self.nerf_mlps_fine = nerf_mlps_fine
        self._template_calls: Dict[Any, F.ProgramCall] = {}
        self._det_u: Dict[Any, torch.Tensor] = {}
        self.precision: Optional[str] = None   # None = package default (functional.set_precision)
        # the constructor's own public attributes shape the compiled programs (hyper_slice_method, use_viewdirs,
        # use_alpha_condition, the frequencies, sample counts ...): changing one later drops the program cache
        self._structural = frozenset(k for k in self.__dict__ if not k.startswith("_") and k != "training")

    def __setattr__(self, name, value):
        # replacing a sub-module (e.g. `model.warp_field = SE3Field(...)`, BASELINE config 5) after a forward pass:
        # the compiled level programs hold the OLD module's parameters — drop them, they are rebuilt on the next call
        # ... and so does any public non-tensor attribute that shapes the programs (hyper_slice_method, use_viewdirs,
        # use_alpha_condition, the frequencies, ...): the cached programs were compiled for the old value
        if "_template_calls" in self.__dict__ and (isinstance(value, nn.Module) or
                                                   name in self.__dict__.get("_structural", ())):
            self._template_calls.clear()
        super().__setattr__(name, value)

    def _live_call(self, key):
        """Cached program for `key`, unless a parameter it was compiled over is no longer a parameter of this model
        (a nested module swapped from outside, e.g. `model.warp_field.mlp = MLP(...)`, which __setattr__ of the
        top-level module never sees): then every cached program is dropped and rebuilt."""
        call = self._template_calls.get(key)
        if call is None:
            return None
        ids = {id(p) for p in self.parameters()}
        if any(id(p) not in ids for p in call.program.params):
            self._template_calls.clear()
            return None
        return call

    # ---- properties of the reference ---------------------------------------------------------
    @property
    def num_nerf_embeds(self):
        return max(self.embeddings_dict[self.nerf_embed_key]) + 1

    @property
    def num_warp_embeds(self):
        return max(self.embeddings_dict[self.warp_embed_key]) + 1

    @property
    def num_hyper_embeds(self):
        return max(self.embeddings_dict[self.hyper_embed_key]) + 1

    @property
    def nerf_embeds(self):
        return torch.tensor(self.embeddings_dict[self.nerf_embed_key])

    @property
    def warp_embeds(self):
        return torch.tensor(self.embeddings_dict[self.warp_embed_key])

    @property
    def hyper_embeds(self):
        return torch.tensor(self.embeddings_dict[self.hyper_embed_key])

    @property
    def has_hyper(self):
        return self.hyper_slice_method != 'none'

    @property
    def has_hyper_embed(self):
        return self.has_hyper

    @property
    def has_embeds(self):
        return self.has_hyper_embed or self.use_warp or self.use_nerf_embed

    def encode_hyper_embed(self, metadata):
        if self.hyper_slice_method in ('axis_aligned_plane', 'bendy_sheet'):
            if self.hyper_use_warp_embed:
                return self.warp_embed(metadata[self.warp_embed_key])
            return self.hyper_embed(metadata[self.hyper_embed_key])
        raise RuntimeError(f'Unknown hyper slice method {self.hyper_slice_method}.')

    def encode_nerf_embed(self, metadata):
        return self.nerf_embed(metadata[self.nerf_embed_key])

    def encode_warp_embed(self, metadata):
        return self.warp_embed(metadata[self.warp_embed_key])

    def apply_warp(self, points, warp_embed, extra_params):
        return self.warp_field(points, self.warp_embed(warp_embed), extra_params)

    # ---- template program ----------------------------------------------------------------------
    def _template_call(self, level: str, n_point_ch: int, xyz_grad: bool, hyper_grad: bool) -> F.ProgramCall:
        """Program of query_template (models.py:447-493): sources 0 = (warped) points (P, 3+H),
        1 = viewdirs (B,3), 2 = template GLO embedding (B,G)."""
        key = (level, n_point_ch, xyz_grad, hyper_grad)
        call = self._live_call(key)
        if call is None:
            m = self.nerf_mlps_fine if level == 'fine' else self.nerf_mlps_coarse
            feats = posenc_features(0, range(3), self.xyz_freq, xyz_grad)
            if n_point_ch > 3:
                feats += posenc_features(0, range(3, n_point_ch), self.hyper_freq, hyper_grad)
            if len(feats) != m.in_ch:
                raise RuntimeError(f"template input has {len(feats)} channels, the MLP expects {m.in_ch} "
                                   "(the reference shape-errors the same way, e.g. axis_aligned_plane needs "
                                   "hyper_slice_out_dim == GLO_dim)")
            rgb_feats = posenc_features(1, range(3), self.dir_freq, False) if self.use_viewdirs else []
            alpha_aux = None
            if self.use_nerf_embed:
                if self.use_alpha_condition:
                    alpha_aux = AuxSpec(copy_features(2, range(self.GLO_dim), True))
                if self.use_rgb_condition:
                    rgb_feats += copy_features(2, range(self.GLO_dim), True)
            layers = modules.nerf_mlp_layers(m, f"nerf_mlps_{level}", AuxSpec(feats), alpha_aux,
                                             AuxSpec(rgb_feats) if rgb_feats else None)
            call = F.ProgramCall(Program(layers, name=f"template_{level}"), [False, True, True], [3, 1],
                                 [("g", 0), ("g", 1), ("y", 0)])
            self._template_calls[key] = call
        return call

    def _gather_table(self, use_warp: bool):
        """(GLOEmbed module, metadata key) when every embedding the TEMPLATE stage of a level consumes — the alpha /
        rgb conditions (models.py:425-436) and axis-aligned hyper coordinates (models.py:533-534) — is a row of ONE
        table looked up with ONE index; None otherwise (or when the stage consumes no embedding at all)."""
        need = []
        if self.use_nerf_embed:
            need.append((self.warp_embed, self.warp_embed_key) if self.hyper_use_warp_embed
                        else (self.nerf_embed, self.nerf_embed_key))
        if use_warp and self.hyper_slice_method == 'axis_aligned_plane':
            need.append((self.warp_embed, self.warp_embed_key) if self.hyper_use_warp_embed
                        else (self.hyper_embed, self.hyper_embed_key))
        if not need or any(n[0] is not need[0][0] or n[1] != need[0][1] for n in need):
            return None
        return need[0]

    def _template_gather_call(self, level: str, hyper_from_table: bool, xyz_grad: bool) -> F.ProgramCall:
        """query_template (models.py:447-493) for levels whose warp runs outside the program (SE3Field, or no warp at
        all): sources 0 = spatial points (P,3), 1 = viewdirs (B,3), 2 = the GLO table, gathered with the ray's index
        by the kernels — conditions and, for the axis-aligned slice, the hyper coordinates are read from the row, and
        the row's gradient is reduced and scattered by the backward machine."""
        key = ("tgather", level, hyper_from_table, xyz_grad)
        call = self._live_call(key)
        if call is None:
            m = self.nerf_mlps_fine if level == 'fine' else self.nerf_mlps_coarse
            G = self.GLO_dim
            feats = posenc_features(0, range(3), self.xyz_freq, xyz_grad)
            if hyper_from_table:
                feats += posenc_features(2, range(G), self.hyper_freq, True)
            if len(feats) != m.in_ch:
                raise RuntimeError(f"template input has {len(feats)} channels, the MLP expects {m.in_ch} "
                                   "(the reference shape-errors the same way, e.g. axis_aligned_plane needs "
                                   "hyper_slice_out_dim == GLO_dim)")
            rgb_feats = posenc_features(1, range(3), self.dir_freq, False) if self.use_viewdirs else []
            alpha_aux = None
            if self.use_nerf_embed:
                if self.use_alpha_condition:
                    alpha_aux = AuxSpec(copy_features(2, range(G), True))
                if self.use_rgb_condition:
                    rgb_feats += copy_features(2, range(G), True)
            layers = modules.nerf_mlp_layers(m, f"nerf_mlps_{level}", AuxSpec(feats), alpha_aux,
                                             AuxSpec(rgb_feats) if rgb_feats else None)
            call = F.ProgramCall(Program(layers, name=f"template_{level}", no_direct=(2,)), [False, True, True], [3, 1],
                                 [("g", 0), ("g", 1), ("y", 0)], gather_src=2)
            self._template_calls[key] = call
        return call

    def _template_reuse_call(self, level: str, n_hyper_pts: int, hyper_from_table: bool, warped_grad: bool) -> F.ProgramCall:
        """query_template (models.py:447-493) over warped points that ALREADY exist — the fine level's pass over the
        coarse level's samples (REUSE_COARSE below).  Sources: 0 = warped points (P, 3 + H) as the coarse level program
        wrote them (xyz in columns 0-2; bendy sheet: the sheet's H outputs behind them), 1 = viewdirs (B,3), 2 = the
        GLO table gathered by ray (conditions; the axis-aligned slice's hyper coordinates).  The gradient w.r.t. source 0
        is what the coarse level program takes as the external gradient on its `warped_points` output."""
        key = ("treuse", level, n_hyper_pts, hyper_from_table, warped_grad)
        call = self._live_call(key)
        if call is None:
            m = self.nerf_mlps_fine if level == 'fine' else self.nerf_mlps_coarse
            G = self.GLO_dim
            feats = posenc_features(0, range(3), self.xyz_freq, warped_grad)
            if n_hyper_pts:
                feats += posenc_features(0, range(3, 3 + n_hyper_pts), self.hyper_freq, warped_grad)
            elif hyper_from_table:
                feats += posenc_features(2, range(G), self.hyper_freq, True)
            if len(feats) != m.in_ch:
                raise RuntimeError(f"template input has {len(feats)} channels, the MLP expects {m.in_ch}")
            rgb_feats = posenc_features(1, range(3), self.dir_freq, False) if self.use_viewdirs else []
            alpha_aux = None
            if self.use_nerf_embed:
                if self.use_alpha_condition:
                    alpha_aux = AuxSpec(copy_features(2, range(G), True))
                if self.use_rgb_condition:
                    rgb_feats += copy_features(2, range(G), True)
            layers = modules.nerf_mlp_layers(m, f"nerf_mlps_{level}", AuxSpec(feats), alpha_aux,
                                             AuxSpec(rgb_feats) if rgb_feats else None)
            call = F.ProgramCall(Program(layers, name=f"template_{level}_reuse", no_direct=(2,)), [False, True, True],
                                 [3, 1], [("g", 0), ("g", 1), ("y", 0)], gather_src=2)
            self._template_calls[key] = call
        return call

    # ---- fused level program -------------------------------------------------------------------
    FUSE_LEVELS = True      # warp field -> hyper sheet -> template as ONE launch per level where the model allows it
    # The fine level's samples are sort(cat(coarse samples, new samples)) (models.py:752-768, model_utils.py:206-232):
    # half (config 3: a third) of its points ARE the coarse level's points, and the warp field / hyper sheet — shared by
    # both levels — have already been evaluated there.  With REUSE_COARSE the fine level runs as two launches: the
    # fine TEMPLATE alone over the coarse level's warped points (read back from HBM, 28 B per point) and the whole
    # level program over the NEW samples only; the compositing kernel reads both through the merge permutation of
    # hn_sample_pdf_split.  Same function values (the warp of a point does not depend on which launch computes it);
    # the fine loss reaches the warp field through the coarse level program's external `warped_points` gradient.
    # At config 2: 1/3 of all warp + sheet forward, backward and weight-gradient work and ~10 % of the stash bytes.
    REUSE_COARSE = os.environ.get("HN_REUSE_COARSE", "1") != "0"
    PREPACK = os.environ.get("HN_PREPACK", "1") != "0"      # one pack launch for the programs of a step (_prepack)

    def _can_fuse_level(self, use_warp: bool, metadata_encoded: bool, metadata) -> bool:
        """One launch per level needs: a TranslationField warp, hyper coordinates from the sheet MLP (or none), and
        ONE GLO table + index for every embedding the level consumes (the only constructible configuration of the
        reference, share_GLO: models.py:167-186), looked up by the kernels themselves."""
        if not (self.FUSE_LEVELS and use_warp and self.use_warp) or metadata_encoded:
            return False
        if not isinstance(self.warp_field, warping.TranslationField):
            return False
        if self.hyper_slice_method not in ('bendy_sheet', 'none', 'axis_aligned_plane'):
            return False
        if self.hyper_slice_method != 'none' and not self.hyper_use_warp_embed:
            return False
        if self.use_nerf_embed and not self.hyper_use_warp_embed:
            return False
        if metadata.get('hyper_point') is not None:
            return False
        if not (self.hyper_sheet_out_dim <= 4 or self.hyper_slice_method != 'bendy_sheet'):
            return False
        # warp, sheet and template share ONE budget of staged source components / source-gradient rows in a fused
        # program (32 each): a configuration that compiles network by network (a large GLO_dim) may not fit — then the
        # level runs unfused instead of raising
        if ("nofuse",) in self._template_calls:
            return False
        try:
            self._level_call('coarse')
        except NotImplementedError:
            self._template_calls[("nofuse",)] = None
            return False
        return True

    def _level_call(self, level: str) -> F.ProgramCall:
        """The whole level as one program of the MLP machine (reference: map_points models.py:545-581 followed by
        query_template models.py:447-493).  Sources: 0 = sample points (P,3), 1 = viewdirs (B,3), 2 = the GLO table
        (gathered with the ray's index), 3 = the warped points — published by the warp / sheet heads in the forward
        launch (they never leave the workgroup), read back from the output tensor in the backward launch.
        Outputs: 0 = warped points (P, 3+H), 1 = rgb (P,3), 2 = alpha (P,1)."""
        key = ("level", level)
        call = self._live_call(key)
        if call is None:
            m = self.nerf_mlps_fine if level == 'fine' else self.nerf_mlps_coarse
            G = self.GLO_dim
            wf = self.warp_field
            layers = modules.mlp_layers(wf.mlp, "warp_field.mlp", wf.input_aux(0, 2, False, True), None,
                                        OutSpec(0, 0, "none", residual=(0, 0), publish=(3, 0)),
                                        GradIn(7, 0, from_dsrc=(3, 0)))
            h = 0
            if self.hyper_slice_method == 'bendy_sheet':
                hs = self.hyper_sheet_mlp
                h = hs.out_ch
                layers += modules.mlp_layers(hs.mlp, "hyper_sheet_mlp.mlp", hs.input_aux(0, 2, False, True), None,
                                             OutSpec(0, 3, "none", publish=(3, 3)), GradIn(7, 3, from_dsrc=(3, 3)))
            feats = posenc_features(3, range(3), self.xyz_freq, True)
            fill = None
            if h:
                feats += posenc_features(3, range(3, 3 + h), self.hyper_freq, True)
            elif self.hyper_slice_method == 'axis_aligned_plane':
                # hyper coordinates = the ray's GLO row itself (models.py:533-534): encoded from the gathered source;
                # the kernel does not copy them into `warped_points`, run_program fills those columns
                h = G
                feats += posenc_features(2, range(G), self.hyper_freq, True)
                fill = (0, 3)
            if len(feats) != m.in_ch:
                raise RuntimeError(f"template input has {len(feats)} channels, the MLP expects {m.in_ch} (the "
                                   "reference shape-errors the same way, e.g. axis_aligned_plane needs "
                                   "hyper_slice_out_dim == GLO_dim)")
            rgb_feats = posenc_features(1, range(3), self.dir_freq, False) if self.use_viewdirs else []
            alpha_aux = None
            if self.use_nerf_embed:
                if self.use_alpha_condition:
                    alpha_aux = AuxSpec(copy_features(2, range(G), True))
                if self.use_rgb_condition:
                    rgb_feats += copy_features(2, range(G), True)
            layers += modules.nerf_mlp_layers(m, f"nerf_mlps_{level}", AuxSpec(feats), alpha_aux,
                                              AuxSpec(rgb_feats) if rgb_feats else None, dst_rgb=1, dst_alpha=2)
            call = F.ProgramCall(Program(layers, name=f"level_{level}", no_direct=(2,)), [False, True, True, False],
                                 [3 + h, 3, 1],
                                 [("g", 1), ("g", 2), ("y", 1), ("go", 0)], gather_src=2, bwd_src_from_out={3: 0},
                                 fill_from_gather=fill)
            self._template_calls[key] = call
        return call

    def compiled_programs(self, n_rays: int, reference: bool = False):
        """[(name, machine.Program, points evaluated per forward pass of `n_rays` rays)] of the programs this model has
        compiled so far (i.e. after a forward pass) — what bench.py prices the MFMA roofline with.  `reference`: the
        point counts of the REFERENCE's forward (every fine sample through every network, models.py:752-768) whatever
        this model executed (REUSE_COARSE evaluates the shared networks on fewer points)."""
        nc, nf = self.num_coarse_samples, self.num_fine_samples
        out = []
        fused = False
        reuse = None if reference else getattr(self, "_reused_coarse", None)
        for key, call in self._template_calls.items():
            if call is None:            # the ("nofuse",) marker
                continue
            if key[0] == "level":
                n_fine = nf if reuse else nc + nf       # with REUSE_COARSE the fine level program sees the new samples only
                out.append((f"level_{key[1]}", call.program, n_rays * (nc if key[1] == 'coarse' else n_fine)))
                fused = True
            elif key[0] == "treuse":
                if reuse and not any(o[0] == f"level_{key[1]}_reuse" for o in out):      # (training / inference builds)
                    out.append((f"level_{key[1]}_reuse", call.program, n_rays * nc))
            else:
                lvl = key[1] if key[0] == "tgather" else key[0]
                out.append((f"template_{lvl}", call.program, n_rays * (nc if lvl == 'coarse' else nc + nf)))
        if fused:
            return [o for o in out if o[0].startswith("level_")]      # incl. level_fine_reuse
        both = n_rays * ((nc + nf) if reuse == 'outside' else (2 * nc + nf))    # REUSE_COARSE: coarse + new samples
        for attr, name in (("warp_field", "warp_field"), ("hyper_sheet_mlp", "hyper_sheet_mlp")):
            mod = getattr(self, attr, None)
            for call in getattr(mod, "_calls", {}).values():
                out.append((name, call.program, both))
        return out

    def get_condition_inputs(self, viewdirs, metadata, metadata_encoded=False):
        """The GLO part of the template conditions (reference: models.py:404-445); view-direction encoding is
        generated inside the template machine."""
        if not self.use_nerf_embed:
            return None
        if metadata_encoded:
            return metadata['encoded_nerf']
        if self.hyper_use_warp_embed:
            return self.warp_embed(metadata[self.warp_embed_key])
        return self.nerf_embed(metadata[self.nerf_embed_key])

    def map_points(self, points, warp_embed, hyper_embed, extra_params, use_warp=True,
                   return_warp_jacobian=False, hyper_point_override=None):
        """points (B,S,3), per-ray embeddings (B,G) -> warped points (B,S,3+H) (reference: models.py:545-581)."""
        if not use_warp:
            return points, None
        if return_warp_jacobian:
            raise NotImplementedError
        if hyper_point_override is not None:
            raise NotImplementedError('hyper_point_override is not implemented.')
        spatial = self.warp_field.warp(points, warp_embed, extra_params) if self.use_warp else points
        if self.hyper_slice_method == 'axis_aligned_plane':
            hyper = hyper_embed[:, None, :].expand(points.shape[0], points.shape[1], hyper_embed.shape[-1])
        elif self.hyper_slice_method == 'bendy_sheet':
            hyper = self.hyper_sheet_mlp(points, hyper_embed)
        else:
            hyper = None
        warped = spatial if hyper is None else torch.cat([spatial, hyper], dim=-1)
        return warped, None

    def render_samples(self, level, points, z_vals, directions, viewdirs, metadata, extra_params, use_warp=True,
                       metadata_encoded=False, return_warp_jacobian=False, use_sample_at_infinity=False,
                       render_opts=None, noise=None, then_pdf=None):
        """One level of the render (reference: models.py:587-671).  `then_pdf` (forward() for the coarse level): the fine
        level's inverse-CDF sampling rides on this level's compositing launch; its results land in out['_pdf']."""
        self._then_pdf = then_pdf
        # filter_sigma (reference models.py:35-63, call site :650) runs inside the compositing kernel: the dust
        # threshold as a scalar, the bounding box as a 0/1 mask over the sample points
        dust, keep = None, None
        if render_opts is not None:
            if 'dust_threshold' in render_opts:
                dust = float(render_opts.get('dust_threshold', 0.0))
            if 'bounding_box' in render_opts:
                xmin, xmax, ymin, ymax, zmin, zmax = render_opts['bounding_box']
                keep = ((points[..., 0] >= xmin) & (points[..., 0] <= xmax) & (points[..., 1] >= ymin)
                        & (points[..., 1] <= ymax) & (points[..., 2] >= zmin) & (points[..., 2] <= zmax)).float()
        b, s = points.shape[0], points.shape[1]
        out = {'points': points}
        if self._can_fuse_level(use_warp, metadata_encoded, metadata):
            if return_warp_jacobian:
                raise NotImplementedError
            idx = metadata[self.warp_embed_key]
            if idx.shape[-1] == 1 and idx.dim() > 1:
                idx = idx.squeeze(-1)
            call = self._level_call(level)
            warped, rgb, alpha = F.run_program(call, [points.reshape(b * s, 3), viewdirs if self.use_viewdirs else None,
                                                      self.warp_embed.embed.weight, None], s, self.precision,
                                               gather_idx=idx)
            warped = warped.view(b, s, -1)
            return self._composite_level(out, warped, rgb, alpha, z_vals, directions, noise, use_sample_at_infinity,
                                         dust, keep, b, s, points.device, level)
        tab = None
        if (self.FUSE_LEVELS and not metadata_encoded and metadata.get('hyper_point') is None
                and not return_warp_jacobian and not (use_warp and self.hyper_slice_method == 'bendy_sheet')):
            tab = self._gather_table(use_warp)
        if tab is not None:
            # a GLO table too wide for the staging / source-gradient budget of ONE program: the generic path below
            # (embedding rows looked up first, surplus copies read directly) takes over instead of raising
            if ("nogather",) in self._template_calls:
                tab = None
            else:
                try:
                    self._template_gather_call(level, use_warp and self.hyper_slice_method == 'axis_aligned_plane',
                                               torch.is_grad_enabled() and use_warp)
                except NotImplementedError:
                    self._template_calls[("nogather",)] = None
                    tab = None
        if tab is not None:
            # the warp (if any) runs as its own program; the template reads conditions / axis-aligned hyper
            # coordinates straight from the GLO table (no gather kernel, no (B,S,H) expand + cat in front of it, the
            # row gradient reduced in the backward machine)
            emb_mod, key = tab
            idx = metadata[key]
            if idx.shape[-1] == 1 and idx.dim() > 1:
                idx = idx.squeeze(-1)
            from_table = use_warp and self.hyper_slice_method == 'axis_aligned_plane'
            xyz, warped_rows = self._warp_outside(points, metadata, idx, emb_mod, from_table, use_warp, extra_params)
            ge = torch.is_grad_enabled() and xyz.requires_grad
            call = self._template_gather_call(level, from_table, ge)
            rgb, alpha = F.run_program(call, [xyz.reshape(b * s, 3), viewdirs if self.use_viewdirs else None,
                                              emb_mod.embed.weight], s, self.precision, gather_idx=idx)
            warped = self._warped_of(xyz, warped_rows, from_table, emb_mod, idx, b, s)
            if use_warp:        # what a fine level re-using these samples needs (REUSE_COARSE)
                self._level_state = {'level': level, 'xyz': xyz, 'rows': warped_rows is not None, 'tab': tab}
            return self._composite_level(out, warped, rgb, alpha, z_vals, directions, noise, use_sample_at_infinity,
                                         dust, keep, b, s, points.device, level)
        if use_warp:
            warp_embed = metadata['encoded_warp'] if metadata_encoded else self.warp_embed(metadata[self.warp_embed_key])
        else:
            warp_embed = None
        if self.has_hyper_embed:
            if metadata_encoded:
                hyper_embed = metadata['encoded_hyper']
            elif self.hyper_use_warp_embed:
                hyper_embed = warp_embed
            else:
                hyper_embed = self.hyper_embed(metadata[self.hyper_embed_key])
        else:
            hyper_embed = None
        warped, _ = self.map_points(points, warp_embed, hyper_embed, extra_params, use_warp=use_warp,
                                    return_warp_jacobian=return_warp_jacobian,
                                    hyper_point_override=metadata.get('hyper_point'))
        nerf_embed = self.get_condition_inputs(viewdirs, metadata, metadata_encoded)
        n_ch = warped.shape[-1]
        ge = torch.is_grad_enabled() and warped.requires_grad
        call = self._template_call(level, n_ch, ge, ge and n_ch > 3)
        rgb, alpha = F.run_program(call, [warped.reshape(b * s, n_ch), viewdirs if self.use_viewdirs else None,
                                          nerf_embed], s, self.precision)
        return self._composite_level(out, warped, rgb, alpha, z_vals, directions, noise, use_sample_at_infinity,
                                     dust, keep, b, s, points.device, level)

    def _warp_outside(self, points, metadata, idx, emb_mod, from_table, use_warp, extra_params):
        """The warp of a level whose warp field runs as its own program (SE3Field, or a TranslationField next to a
        gathered template): (xyz (B,S,3), `warped_points` rows (B,S,3+H) written by the exp-map launch | None)."""
        if use_warp and from_table and isinstance(self.warp_field, warping.SE3Field):
            # config 5: the exp-map launch writes `warped_points` = [xyz | GLO row] itself (no index_select + cat)
            return self.warp_field.warp_with_rows(points, emb_mod.embed.weight, idx)
        if use_warp:
            needs_rows = isinstance(self.warp_field, warping.TranslationField)
            return self.warp_field.warp(points, self.warp_embed(metadata[self.warp_embed_key]) if needs_rows else None,
                                        extra_params), None
        return points, None

    @staticmethod
    def _warped_of(xyz, warped_rows, from_table, emb_mod, idx, b, s):
        """`warped_points` of a level whose template read its hyper coordinates from the GLO table."""
        if warped_rows is not None:
            return warped_rows
        if not from_table:
            return xyz
        with torch.no_grad():
            flat = idx.reshape(-1)
            safe = flat.clamp(0, emb_mod.embed.weight.shape[0] - 1)
            rows = emb_mod.embed.weight.index_select(0, safe)
            rows.masked_fill_((safe != flat)[:, None], float("nan"))   # as the kernels' own gather poisons it
        return torch.cat([xyz, rows[:, None, :].expand(b, s, rows.shape[-1])], dim=-1)

    def _can_reuse_coarse(self, use_warp, metadata_encoded, metadata, return_warp_jacobian) -> Optional[str]:
        """'fused' / 'outside' when the fine level may take the warp of the coarse level's samples from the coarse
        level (REUSE_COARSE), else None: the coarse level ran as one fused program, or with its warp field as a program
        of its own in front of a gathered template (SE3Field: BASELINE config 5)."""
        if not (self.REUSE_COARSE and use_warp) or return_warp_jacobian:
            return None
        if self._can_fuse_level(use_warp, metadata_encoded, metadata):
            return 'fused'
        st = getattr(self, '_level_state', None)
        if st is not None and st['level'] == 'coarse' and st['xyz'].shape[1] == self.num_coarse_samples:
            return 'outside'
        return None

    def _prepack(self, use_warp, metadata_encoded, metadata, return_warp_jacobian, device, collect=False):
        """The weight streams of every program this forward pass is about to launch, packed by ONE launch
        (machine.pack_many) where more than one of them is stale — after an optimizer step that is all three of a
        training step's (coarse level, fine level over the new samples, fine template over the coarse samples): three
        dispatches of ~9 us for ~4 us of work each.  A prediction only: a program it misses packs itself as before.
        `collect`: return the pack groups (machine.collect_pack_jobs) instead of launching them — the step head
        (functional.render_prologue) packs them in its own launch."""
        if not self.PREPACK or not self._can_fuse_level(use_warp, metadata_encoded, metadata):
            return []
        from .. import machine
        calls = [self._level_call('coarse')]
        grad = torch.is_grad_enabled() and any(p.requires_grad for p in calls[0].program.params)
        if self.num_fine_samples > 0:
            calls.append(self._level_call('fine'))
            if self.REUSE_COARSE and not return_warp_jacobian:
                n_hyper = self.hyper_sheet_out_dim if self.hyper_slice_method == 'bendy_sheet' else 0
                calls.append(self._template_reuse_call('fine', n_hyper, self.hyper_slice_method == 'axis_aligned_plane', grad))
        if collect:
            groups = machine.collect_pack_jobs([c.runner for c in calls], device, F.mode_of(self.precision), force=grad,
                                               min_jobs=1)
            if len(groups) <= 1:
                return groups
            for m, arr, grp in groups:           # two numeric modes in one model: not the step head's business
                L.launch("hn_pack_units_multi", C.c_int(m), arr, C.c_int(len(grp)), L.stream_handle())
                machine.mark_packed(grp)
            return []
        machine.pack_many([c.runner for c in calls], device, F.mode_of(self.precision), force=grad)
        return []

    def _render_fine_reusing_coarse(self, coarse, points, z_vals, pts_new, perm, directions, viewdirs, metadata,
                                    use_sample_at_infinity, render_opts, noise, extra_params=None, how='fused'):
        """The fine level (models.py:752-768 -> render_samples 587-671) without re-evaluating the shared networks on the
        coarse level's samples (REUSE_COARSE).  `points` / `z_vals` (B, Nc+Nf, .) sorted, `pts_new` (B, Nf, 3) the new
        samples in draw order, `perm` (B, Nc+Nf) the merge permutation.  Returns the same dict as render_samples."""
        b, s = z_vals.shape
        nc = self.num_coarse_samples
        nf = s - nc
        dust, keep = None, None
        if render_opts is not None:
            if 'dust_threshold' in render_opts:
                dust = float(render_opts.get('dust_threshold', 0.0))
            if 'bounding_box' in render_opts:
                xmin, xmax, ymin, ymax, zmin, zmax = render_opts['bounding_box']
                keep = ((points[..., 0] >= xmin) & (points[..., 0] <= xmax) & (points[..., 1] >= ymin)
                        & (points[..., 1] <= ymax) & (points[..., 2] >= zmin) & (points[..., 2] <= zmax)).float()
        vd = viewdirs if self.use_viewdirs else None
        from_table = self.hyper_slice_method == 'axis_aligned_plane'
        w_old = coarse['warped_points']                                   # (B, Nc, 3 + H)
        if how == 'fused':
            idx = metadata[self.warp_embed_key]
            if idx.shape[-1] == 1 and idx.dim() > 1:
                idx = idx.squeeze(-1)
            table = self.warp_embed.embed.weight
            # (i) the coarse level's samples: their warped points exist (output of the coarse program) — the fine
            # template alone
            n_hyper = w_old.shape[-1] - 3 if self.hyper_slice_method == 'bendy_sheet' else 0
            ge = torch.is_grad_enabled() and w_old.requires_grad
            call_old = self._template_reuse_call('fine', n_hyper, from_table, ge)
            rgb_old, alpha_old = F.run_program(call_old, [w_old.reshape(b * nc, w_old.shape[-1]), vd, table], nc,
                                               self.precision, gather_idx=idx)
            # (ii) the new samples: warp field -> hyper sheet -> fine template, one launch
            call_new = self._level_call('fine')
            w_new, rgb_new, alpha_new = F.run_program(call_new, [pts_new.reshape(b * nf, 3), vd, table, None], nf,
                                                      self.precision, gather_idx=idx)
        else:
            # the warp field is a program of its own (SE3Field + exp-map launch): the fine template over the coarse
            # level's warped xyz, then field + template over the new samples
            st = self._level_state
            emb_mod, key = st['tab']
            idx = metadata[key]
            if idx.shape[-1] == 1 and idx.dim() > 1:
                idx = idx.squeeze(-1)
            table = emb_mod.embed.weight
            xyz_old = st['xyz']
            ge = torch.is_grad_enabled() and xyz_old.requires_grad
            rgb_old, alpha_old = F.run_program(self._template_gather_call('fine', from_table, ge),
                                               [xyz_old.reshape(b * nc, 3), vd, table], nc, self.precision, gather_idx=idx)
            xyz_new, rows_new = self._warp_outside(pts_new, metadata, idx, emb_mod, from_table, True, extra_params)
            ge = torch.is_grad_enabled() and xyz_new.requires_grad
            rgb_new, alpha_new = F.run_program(self._template_gather_call('fine', from_table, ge),
                                               [xyz_new.reshape(b * nf, 3), vd, table], nf, self.precision, gather_idx=idx)
            w_new = self._warped_of(xyz_new, rows_new, from_table, emb_mod, idx, b, nf)
            self._level_state = None
        scale = 1.0
        if noise is None and (self.noise_std is not None) and self.noise_std > 0.0 and self.use_stratified_sampling:
            noise = getattr(self, '_auto_noise', {}).pop('fine', None)
            if noise is None or tuple(noise.shape) != (b, s, 1):
                noise = torch.randn((b, s, 1), device=z_vals.device, dtype=torch.float32)
            scale = float(self.noise_std)
        res = F.composite(rgb_old.view(b, nc, 3), alpha_old.view(b, nc), noise, z_vals, directions, w_old, variant=0,
                          white_bg=self.use_white_background, sample_at_infinity=use_sample_at_infinity,
                          want_median=True, dust_threshold=dust, keep=keep, noise_scale=scale,
                          rgb1=rgb_new.view(b, nf, 3), raw1=alpha_new.view(b, nf), warped1=w_new.view(b, nf, -1), perm=perm)
        return {'points': points, 'warped_points': res[6], 'rgb': res[0], 'depth': res[1], 'acc': res[2],
                'weights': res[3], 'med_depth': res[4], 'med_points': res[5].view(b, 1, 1)}

    def _empty_result(self, like: torch.Tensor, use_warp: bool):
        """Zero rays in, zero rays out (the reference's ATen ops all accept empty batches): no kernel is launched."""
        e = lambda *shape: like.new_zeros(shape)
        hyper = 0
        if use_warp and self.hyper_slice_method == 'bendy_sheet':
            hyper = self.hyper_sheet_out_dim
        elif use_warp and self.hyper_slice_method == 'axis_aligned_plane':
            hyper = self.GLO_dim
        out = {}
        for level, s in (('coarse', self.num_coarse_samples), ('fine', self.num_coarse_samples + self.num_fine_samples)):
            if level == 'fine' and self.num_fine_samples <= 0:
                break
            out[level] = {'points': e(0, s, 3), 'warped_points': e(0, s, 3 + hyper), 'rgb': e(0, 3), 'depth': e(0),
                          'acc': e(0), 'weights': e(0, s), 'med_depth': e(0), 'med_points': e(0, 1, 1)}
        return out

    def _composite_level(self, out, warped, rgb, alpha, z_vals, directions, noise, use_sample_at_infinity, dust, keep,
                         b, s, device, level=None):
        """noise_regularize + Softplus + filter_sigma + volumetric_rendering + median-depth gather
        (models.py:485-489, 650-669) in the compositing kernel."""
        scale = 1.0
        if noise is None and (self.noise_std is not None) and self.noise_std > 0.0 and self.use_stratified_sampling:
            noise = getattr(self, '_auto_noise', {}).pop(level, None) if level is not None else None   # drawn by forward()
            if noise is None or tuple(noise.shape) != (b, s, 1):
                noise = torch.randn((b, s, 1), device=device, dtype=torch.float32)
            scale = float(self.noise_std)                                              # scaled inside the kernel
        then_pdf = getattr(self, '_then_pdf', None)
        self._then_pdf = None
        res = F.composite(rgb.view(b, s, 3), alpha.view(b, s), noise, z_vals, directions, warped, variant=0,
                          white_bg=self.use_white_background, sample_at_infinity=use_sample_at_infinity,
                          want_median=True, dust_threshold=dust, keep=keep, noise_scale=scale, then_pdf=then_pdf)
        if then_pdf is not None:
            out['_pdf'] = res[-(6 if then_pdf.get('split') else 4):]
        out['warped_points'] = warped
        out['rgb'], out['depth'], out['acc'], out['weights'], out['med_depth'] = res[0], res[1], res[2], res[3], res[4]
        out['med_points'] = res[5].view(b, 1, 1)
        return out

    def forward(self, rays_dict: Dict[str, Any], extra_params: Dict[str, Any], metadata_encoded=False,
                use_warp=True, return_points=False, return_weights=False, return_warp_jacobian=False, near=None,
                far=None, use_sample_at_infinity=None, render_opts=None, deterministic=False,
                rng: Optional[Dict[str, torch.Tensor]] = None):
        """Returns {'coarse': {...}, 'fine': {...}} with keys points, warped_points, rgb, depth, med_depth, acc,
        weights, med_points (reference: models.py:673-780).  `rng` optionally supplies the random draws
        ('t_rand' (B,Nc), 'u' (B,Nf), 'noise_coarse' (B,Nc,1), 'noise_fine' (B,Nc+Nf,1), noise already scaled);
        otherwise they are drawn from torch's generator in the reference's order."""
        rng = rng or {}
        use_warp = self.use_warp and use_warp
        origins, directions, metadata = rays_dict['origins'], rays_dict['directions'], rays_dict['metadata']
        L.require_gpu(origins, directions)
        viewdirs = rays_dict['viewdirs'] if rays_dict.get('viewdirs') is not None else directions
        near = self.near if near is None else near
        far = self.far if far is None else far
        if use_sample_at_infinity is None:
            use_sample_at_infinity = self.use_sample_at_infinity
        b = origins.shape[0]
        if b == 0:
            return self._empty_result(origins, use_warp)
        # every draw the caller did not supply, in ONE launch (F.random_draws) instead of the reference's four ATen
        # calls (model_utils.py:31 t_rand, :226 u, :300-317 the density noise of each level); the noise stays N(0,1)
        # and is scaled by noise_std inside the compositing kernel
        self._auto_noise = {}
        want = []
        if self.use_stratified_sampling and F.FAST_DRAWS:
            nc, nf = self.num_coarse_samples, self.num_fine_samples
            noisy = (self.noise_std is not None) and self.noise_std > 0.0
            if 't_rand' not in rng:
                want.append(('t_rand', (b, nc), 'uniform'))
            if noisy and 'noise_coarse' not in rng:
                want.append(('noise_coarse', (b, nc, 1), 'normal'))
            if nf > 0 and 'u' not in rng:
                want.append(('u', (b, nf), 'uniform'))
            if nf > 0 and noisy and 'noise_fine' not in rng:
                want.append(('noise_fine', (b, nc + nf, 1), 'normal'))
        self._level_state = None
        lazy_ids = isinstance(metadata, model_utils.RayMetadata) and not metadata.converted()
        z_vals = points = None
        if F.PROLOGUE and (want or lazy_ids):
            # the step head as ONE launch (hn_render_prologue): the draws, the stale weight streams of the programs this
            # pass will run, the coarse samples placed from the t_rand draw, the int64 image ids of the ray rows — four
            # launches of 5-13 us (each ~5 us of dispatch floor) otherwise
            ids = None
            if lazy_ids:
                ids = (metadata.raw, torch.empty(b, dtype=torch.int64, device=origins.device))
            sample = None
            if want and want[0][0] == 't_rand':
                lower, upper, _ = model_utils.stratified_bounds(self.num_coarse_samples, near, far,
                                                                self.use_linear_disparity, origins.device)
                sample = dict(draw=0, origins=origins, directions=directions, lower=lower, upper=upper)
            if ids is not None:
                metadata.set_converted(ids[1])       # (the launch below fills it; nothing reads it before)
            groups = self._prepack(use_warp, metadata_encoded, metadata, return_warp_jacobian, origins.device, collect=True)
            got, z_vals, points = F.render_prologue(origins.device, F.mode_of(self.precision), groups,
                                                    [(shape, kind) for _, shape, kind in want], sample, ids)
        else:
            got = F.random_draws([(shape, kind) for _, shape, kind in want], origins.device) if want else []
            self._prepack(use_warp, metadata_encoded, metadata, return_warp_jacobian, origins.device)
        if want:
            rng = dict(rng)
            for (name, _, _), t in zip(want, got):
                if name.startswith('noise_'):
                    self._auto_noise[name[6:]] = t
                else:
                    rng[name] = t
        if z_vals is None:
            z_vals, points = model_utils.sample_along_rays(origins, directions, self.num_coarse_samples, near, far,
                                                           self.use_stratified_sampling, self.use_linear_disparity,
                                                           t_rand=rng.get('t_rand'))
        u = then_pdf = None
        if self.num_fine_samples > 0:
            u = rng.get('u')
            if u is None:
                if self.use_stratified_sampling:
                    u = torch.rand(b, self.num_fine_samples, device=origins.device)
                else:
                    # the reference's deterministic draws (model_utils.py:226-227): built on the host with the same
                    # ATen CPU op, uploaded once per (batch, device) — never inside a stream capture
                    key = (b, self.num_fine_samples, str(origins.device))
                    u = self._det_u.get(key)
                    if u is None:
                        u = torch.linspace(0, 1, self.num_fine_samples).to(origins.device).expand(b, -1).contiguous()
                        self._det_u[key] = u
            if F.COMPOSITE_PDF and 3 <= self.num_coarse_samples <= 256 and self.num_coarse_samples + self.num_fine_samples <= 512:
                # the fine level's inverse-CDF sampling rides on the coarse level's compositing launch (whether the fine
                # level will re-use the coarse warp is only known after the coarse level ran: the split outputs — merge
                # permutation, new points — are produced whenever REUSE_COARSE could apply)
                then_pdf = dict(u=u, origins=origins, directions=directions,
                                split=bool(self.REUSE_COARSE and use_warp and not return_warp_jacobian))
        coarse = self.render_samples('coarse', points, z_vals, directions, viewdirs, metadata, extra_params,
                                     use_warp=use_warp, metadata_encoded=metadata_encoded,
                                     return_warp_jacobian=return_warp_jacobian,
                                     use_sample_at_infinity=self.use_sample_at_infinity,
                                     noise=rng.get('noise_coarse'), then_pdf=then_pdf)
        pdf = coarse.pop('_pdf', None)
        self._then_pdf = None
        out = {'coarse': coarse}
        if self.num_fine_samples > 0:
            how = self._can_reuse_coarse(use_warp, metadata_encoded, metadata, return_warp_jacobian)
            self._reused_coarse = how
            if how is not None and (pdf is None or len(pdf) == 6):
                if pdf is not None:
                    z_fine, pts_fine, inds, _, perm, pts_new = pdf
                else:
                    z_fine, pts_fine, inds, _, perm, pts_new = F.sample_pdf(coarse['weights'], z_vals, u, origins,
                                                                            directions, split=True)
                fine = self._render_fine_reusing_coarse(coarse, pts_fine, z_fine, pts_new, perm, directions, viewdirs,
                                                        metadata, use_sample_at_infinity, render_opts,
                                                        rng.get('noise_fine'), extra_params, how)
            else:
                if pdf is not None:
                    z_fine, pts_fine, inds = pdf[0], pdf[1], pdf[2]
                else:
                    z_fine, pts_fine, inds, _ = F.sample_pdf(coarse['weights'], z_vals, u, origins, directions)
                fine = self.render_samples('fine', pts_fine, z_fine, directions, viewdirs, metadata, extra_params,
                                           use_warp=use_warp, metadata_encoded=metadata_encoded,
                                           return_warp_jacobian=return_warp_jacobian,
                                           use_sample_at_infinity=use_sample_at_infinity, render_opts=render_opts,
                                           noise=rng.get('noise_fine'))
            out['fine'] = fine
            # not part of the reference's return value: kept for tests / debugging
            self.last_sampling = {'z_coarse': z_vals, 'z_fine': z_fine, 'inds': inds, 'u': u}
        self._level_state = None        # (holds the coarse level's warped xyz, i.e. its autograd graph)
        return out
