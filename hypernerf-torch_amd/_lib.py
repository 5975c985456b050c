"""ctypes binding of the C-ABI shared library (include/hn_kernels.h).

The product path has NO CPU fallback: if the library is missing, or a tensor is not on a
ROCm device, the ops raise.  `load()` is lazy so that module construction, state_dict
handling and program building work on a machine without a GPU (tests -m "not gpu").
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
PRODUCT_LIB_PATH = os.path.join(CSRC, "libhn_hip.so")
# HN_LIB_PATH: run under ANOTHER build of the library (the A/B tools compare prebuilt variants without ever overwriting
# the product library); build() / needs_build() only ever write the product path
LIB_PATH = os.environ.get("HN_LIB_PATH") or PRODUCT_LIB_PATH
SOURCES = ["hn_mlp.hip", "hn_render.hip", "hn_calib.hip"]
CSRC_HEADERS = ["hn_common.h", "hn_pack.h"]
HEADER = os.path.join(os.path.dirname(_HERE), "include", "hn_kernels.h")
BUILD_MACROS = ("HN_WGRAD_PERSIST", "HN_REDUCE_SPLIT", "HN_BF16_WAVES", "HN_PROF", "HN_CHUNK_UNITS", "HN_WGRAD_AUX", "HN_WGRAD_EXP", "HN_WGRAD_BIAS_MFMA", "HN_WGRAD_BLOCK", "HN_WGRAD_STAGES", "HN_WGRAD_MAXSLOT", "HN_WSTREAM_ASYM")     # build-time tuning knobs (A/B experiments)

HN_MODE_F32, HN_MODE_BF16, HN_MODE_BF16_S8 = 0, 1, 2
BUILD_CONFIG_KEYS = ("WGRAD_STAGES", "WGRAD_MAXSLOT", "WGRAD_BIAS_MFMA", "CHUNK_UNITS", "WGRAD_BLOCK", "WSTREAM_ASYM",
                     "BF16_WAVES", "WGRAD_AUX")


def _read_build_config(path: str):
    """The build-time knobs of the library at `path`, or None when there is no library (or one older than ABI 340).
    Read from the marker string the library carries (`hn_build_config_text`) WITHOUT dlopen-ing it: this runs at import
    time, and a library that was dlopen-ed once stays mapped under its path — a rebuild that follows (build(force=True))
    would then never be seen by load().  load() cross-checks the text against hn_build_config() of the mapped code."""
    import re
    try:
        with open(path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    m = re.search(rb"HN_BUILD_CONFIG:([0-9, ]+);", blob)
    if m is None:
        return None
    vals = [int(x) for x in m.group(1).decode().replace(" ", "").split(",")]
    return {k: v for k, v in zip(BUILD_CONFIG_KEYS, vals)}


# hn_wgrad_kernel's LDS ring, the weight-stream chunk and the bias path of the build: host-side mirrors of build-time
# macros in csrc/hn_mlp.hip.  They come from the LIBRARY (hn_build_config) whenever one exists — a library prebuilt
# with other knobs carries its own values; the environment variables (which also drive build()) only stand in before the
# first build, and load() refuses a library that disagrees with the mirrors in use.
BUILD_CONFIG = _read_build_config(LIB_PATH)
_cfg = BUILD_CONFIG or {}
WGRAD_STAGES = _cfg.get("WGRAD_STAGES", int(os.environ.get("HN_WGRAD_STAGES") or 2))
WGRAD_MAXSLOT = _cfg.get("WGRAD_MAXSLOT", int(os.environ.get("HN_WGRAD_MAXSLOT") or 8))
WGRAD_BIAS_MFMA = bool(_cfg.get("WGRAD_BIAS_MFMA", os.environ.get("HN_WGRAD_BIAS_MFMA", "0") not in ("", "0")))
WGRAD_MAX_STAGE_KB = min(8 * WGRAD_MAXSLOT, 160 // WGRAD_STAGES)
HN_MAX_SRC, HN_MAX_DST, HN_MAX_SLOTS = 8, 4, 128
HN_OP_WORDS, HN_CHUNK_UNITS, HN_DSRC_COMPS = 8, _cfg.get("CHUNK_UNITS", int(os.environ.get("HN_CHUNK_UNITS", 32))), 32
HN_AUXG_MAX = 3
HN_MAX_COMPS = 32

HN_OP_LAYER, HN_OP_OUT, HN_OP_OUT_WIDE = 1, 4, 5
HN_ACT_NONE, HN_ACT_RELU = 0, 1
HN_LAYER_NO_COMMIT, HN_LAYER_DIRECT = 1, 2
HN_BOP_LOAD, HN_BOP_LOAD_WIDE, HN_BOP_LAYER, HN_BOP_AUX = 1, 2, 3, 4
HN_FEAT_ZERO, HN_FEAT_ID, HN_FEAT_SIN, HN_FEAT_COS, HN_FEAT_SINP, HN_FEAT_ID_DIRECT = 0, 1, 2, 3, 4, 5


class HnSrc(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ld", C.c_int32), ("per_ray", C.c_int32), ("gather_idx", C.c_void_p),
                ("gather_rows", C.c_int32), ("pad", C.c_int32)]


class HnDst(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("ld", C.c_int32), ("pad", C.c_int32)]


class HnSlot(C.Structure):
    _fields_ = [("off", C.c_uint64), ("nt", C.c_int32), ("pad", C.c_int32)]


class HnMlpArgs(C.Structure):
    _fields_ = [
        ("mode", C.c_int32), ("n_points", C.c_int32), ("samples_per_ray", C.c_int32), ("training", C.c_int32),
        ("n_ops", C.c_int32), ("n_chunks", C.c_int32), ("n_dsrc", C.c_int32), ("n_bias", C.c_int32),
        ("n_feat", C.c_int32), ("max_groups", C.c_int32),
        ("ops", C.c_void_p), ("wstream", C.c_void_p), ("bias", C.c_void_p), ("feat", C.c_void_p),
        ("stash", C.c_void_p), ("masks", C.c_void_p), ("dsrc", C.c_void_p),
        ("src", HnSrc * HN_MAX_SRC), ("dst", HnDst * HN_MAX_DST), ("slots", HnSlot * HN_MAX_SLOTS),
        ("prof", C.c_void_p), ("comps", C.c_void_p), ("n_comps", C.c_int32), ("embed_reg_mask", C.c_int32),
        ("embed_grad", C.c_void_p), ("embed_idx", C.c_void_p), ("embed_rows", C.c_int32), ("embed_dim", C.c_int32),
        ("embed_col", C.c_int8 * 32), ("n_trig_comps", C.c_int32), ("wide_ops", C.c_int32), ("dz_scale_log2", C.c_int32), ("trig_lo_planes", C.c_int32),
        ("timeline", C.c_void_p), ("embed_partial", C.c_void_p),
    ]


class HnDwBatch(C.Structure):
    _fields_ = [("jobs", C.c_void_p), ("stash", C.c_void_p), ("grads", C.c_void_p), ("n_jobs", C.c_int32),
                ("pad", C.c_int32), ("partials", C.c_void_p)]


HN_MAX_WGRAD_BATCH = 8
HN_MAX_DRAWS = 8


HN_MAX_PACK_JOBS = 8


class HnPackJob(C.Structure):
    _fields_ = [("units", C.c_void_p), ("ptrs", C.c_void_p), ("wstream", C.c_void_p), ("bias", C.c_void_p),
                ("bias_out", C.c_void_p), ("n_units", C.c_int32), ("n_bias", C.c_int32)]


class HnEmbedReduce(C.Structure):
    _fields_ = [("grad", C.c_void_p), ("rows", C.c_int32), ("dim", C.c_int32), ("col_mask", C.c_uint32), ("n_src", C.c_int32),
                ("partial", C.c_void_p * HN_MAX_WGRAD_BATCH), ("idx", C.c_void_p * HN_MAX_WGRAD_BATCH),
                ("n_blocks", C.c_int32 * HN_MAX_WGRAD_BATCH), ("samples_per_ray", C.c_int32 * HN_MAX_WGRAD_BATCH)]


class HnDraw(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("n", C.c_int64), ("kind", C.c_int32), ("pad", C.c_int32)]


class HnPrologue(C.Structure):
    _fields_ = [("t_rand_draw", C.c_int32), ("n_rays", C.c_int32), ("n", C.c_int32), ("ray_ld", C.c_int32),
                ("per_ray_bounds", C.c_int32), ("scale", C.c_float),
                ("origins", C.c_void_p), ("dirs", C.c_void_p), ("lower", C.c_void_p), ("upper", C.c_void_p),
                ("z_out", C.c_void_p), ("pts_out", C.c_void_p), ("ids_src", C.c_void_p), ("ids_dst", C.c_void_p),
                ("ids_ld", C.c_int32), ("n_ids", C.c_int32)]


class HnCompositeArgs(C.Structure):
    _fields_ = [
        ("variant", C.c_int32), ("n_rays", C.c_int32), ("n_samples", C.c_int32), ("white_bg", C.c_int32),
        ("sample_at_infinity", C.c_int32), ("warped_ld", C.c_int32), ("has_dust", C.c_int32), ("dust_threshold", C.c_float),
        ("rgb", C.c_void_p), ("raw", C.c_void_p), ("noise", C.c_void_p), ("z", C.c_void_p), ("dirs", C.c_void_p),
        ("ray_ld", C.c_int64), ("warped", C.c_void_p),
        ("out_rgb", C.c_void_p), ("out_depth", C.c_void_p), ("out_acc", C.c_void_p), ("out_weights", C.c_void_p),
        ("out_med_depth", C.c_void_p), ("out_med_points", C.c_void_p),
        ("g_rgb", C.c_void_p), ("g_depth", C.c_void_p), ("g_acc", C.c_void_p), ("g_weights", C.c_void_p),
        ("d_rgb", C.c_void_p), ("d_raw", C.c_void_p), ("keep", C.c_void_p),
        ("noise_scale", C.c_float), ("split", C.c_int32),
        ("perm", C.c_void_p), ("rgb1", C.c_void_p), ("raw1", C.c_void_p), ("warped1", C.c_void_p),
        ("d_rgb1", C.c_void_p), ("d_raw1", C.c_void_p), ("out_warped", C.c_void_p),
    ]


# numpy mirrors of the table structs (uploaded to the device as raw bytes)
PACK_UNIT_DT = np.dtype([("w_id", "<i4"), ("ld", "<i4"), ("r0", "<i4"), ("c0", "<i4"), ("r_end", "<i4"),
                         ("c_end", "<i4"), ("k0", "<i4"), ("transposed", "<i4")])
PACK_BIAS_DT = np.dtype([("w_id", "<i4"), ("n", "<i4"), ("off", "<i4"), ("len", "<i4")])
FEAT_DT = np.dtype([("packed", "<i4"), ("freq", "<f4")])
DWJOB_DT = np.dtype([("z_off", "<u8"), ("x_off", "<u8"), ("z_nt", "<i4"), ("x_nt", "<i4"), ("z_t0", "<i4"),
                     ("x_t0", "<i4"), ("n_nt", "<i4"), ("n_kt", "<i4"), ("blk0", "<i4"), ("blk1", "<i4"),
                     ("w_off", "<i4"), ("ld", "<i4"), ("r0", "<i4"), ("c0", "<i4"), ("r_end", "<i4"),
                     ("c_end", "<i4"), ("b_off", "<i4"), ("pad", "<i4"), ("x2_off", "<u8"), ("x2_nt", "<i4"),
                     ("x2_t0", "<i4"), ("n_kt1", "<i4"), ("p_tile", "<i4")])
DWREDUCE_DT = np.dtype([("batch", "<i4"), ("w_off", "<i4"), ("ld", "<i4"), ("row0", "<i4"), ("col0", "<i4"),
                        ("r_end", "<i4"), ("c_end", "<i4"), ("first", "<i4"), ("count", "<i4")])

ADAM_RANGE_DT = np.dtype([("start", "<i8"), ("len", "<i4"), ("pad", "<i4")])


class HnAdamFuse(C.Structure):
    _fields_ = [("params", C.c_void_p), ("grads", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("n", C.c_int64), ("hyper", C.c_void_p), ("step", C.c_void_p), ("rest", C.c_void_p),
                ("n_rest", C.c_int32), ("zero_grad", C.c_int32)]


EXPORTS = ["hn_version", "hn_abi_sizes", "hn_build_config", "hn_mlp_wgrad_reduce_adam", "hn_render_prologue", "hn_pack_units", "hn_pack_units_multi", "hn_mlp_forward", "hn_mlp_backward", "hn_mlp_wgrad",
           "hn_mlp_wgrad_batched", "hn_mlp_wgrad_batched_t", "hn_mlp_wgrad_reduce", "hn_mlp_workspace_bytes",
           "hn_sample_along_rays", "hn_sample_legacy", "hn_posenc", "hn_composite_forward", "hn_composite_backward", "hn_sample_pdf", "hn_sample_pdf_split", "hn_composite_sample_pdf",
           "hn_embed_gather", "hn_embed_backward", "hn_se3_apply_forward", "hn_se3_apply_backward", "hn_se3_warp_forward", "hn_se3_warp_backward", "hn_generate_rays", "hn_adam_step",
           "hn_mse_loss_forward", "hn_mse_loss_backward", "hn_mse_loss_forward_grad", "hn_depth_index", "hn_random_fill",
           "hn_probe_mfma", "hn_calib_mfma", "hn_calib_stream", "hn_calib_stream_pattern", "hn_calib_ring"]

_lib = None


class HnError(RuntimeError):
    pass


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build() -> bool:
    if not os.path.exists(PRODUCT_LIB_PATH):
        return True
    t = os.path.getmtime(PRODUCT_LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, h) for h in CSRC_HEADERS] + [HEADER]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP sources for gfx950 into csrc/libhn_hip.so (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return PRODUCT_LIB_PATH
    out = os.environ.get("HN_BUILD_OUT") or PRODUCT_LIB_PATH      # HN_BUILD_OUT: an A/B variant beside the product library
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "-fPIC", "-shared",
           "-o", out] + [os.path.join(CSRC, s) for s in SOURCES]
    for macro in BUILD_MACROS:          # build-time tuning knobs (A/B experiments)
        if os.environ.get(macro):
            cmd.insert(1, f"-D{macro}={os.environ[macro]}")
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise HnError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stdout + res.stderr)
    if verbose:
        print(res.stdout + res.stderr)
    return out


def build_id() -> dict:
    """Identity of the kernels a measurement was taken on: sha256 over the kernel sources (+ the build-time tuning
    macros in the environment) and over the built library.  bench.py prints it; tools/make_profiles.py stamps the PMC
    traffic summaries with it, and bench.py only quotes a summary whose source hash equals the running build's."""
    import hashlib
    h = hashlib.sha256()
    for d in [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, h) for h in CSRC_HEADERS] + [HEADER]:
        with open(d, "rb") as f:
            h.update(f.read())
    for macro in BUILD_MACROS:
        h.update(f"{macro}={os.environ.get(macro, '')};".encode())
    lib = None
    if os.path.exists(LIB_PATH):
        with open(LIB_PATH, "rb") as f:
            lib = hashlib.sha256(f.read()).hexdigest()[:16]
    ident = {"kernel_src_sha256": h.hexdigest()[:16], "lib_sha256": lib}
    if LIB_PATH != PRODUCT_LIB_PATH:      # an A/B variant: the source hash above describes the tree, not this library
        ident["lib_path_override"] = LIB_PATH
        ident["kernel_src_sha256"] = "override:" + (lib or "?")
    if BUILD_CONFIG is not None:
        ident["build_config"] = dict(BUILD_CONFIG)
    return ident


def load():
    """dlopen the library (building it first only if hipcc is around and sources are newer)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HnError(
            f"{LIB_PATH} is missing: the HIP extension has not been built (run __graft_entry__.build()). "
            "hypernerf_torch_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise HnError(f"{LIB_PATH} does not export {name}")
        getattr(lib, name).restype = C.c_int
    # the host tables in use were cut for the mirrors above: a library built with other knobs must not run under them
    # (e.g. bias records whose slab tiles a bias-by-MFMA build never writes: garbage gradients, no error)
    buf = (C.c_int32 * len(BUILD_CONFIG_KEYS))()
    n_cfg = lib.hn_build_config(buf, len(BUILD_CONFIG_KEYS))
    now = {k: int(buf[i]) for i, k in enumerate(BUILD_CONFIG_KEYS[:n_cfg])}
    mine = {"WGRAD_STAGES": WGRAD_STAGES, "WGRAD_MAXSLOT": WGRAD_MAXSLOT, "WGRAD_BIAS_MFMA": int(WGRAD_BIAS_MFMA),
            "CHUNK_UNITS": HN_CHUNK_UNITS}
    bad = {k: (v, now.get(k)) for k, v in mine.items() if now.get(k) != v}
    if bad:
        raise HnError(f"{LIB_PATH} was built with other tuning knobs than the host tables assume (mirror, library): {bad}; "
                      "re-import hypernerf_torch_amd after building, or unset the HN_* build variables")
    _lib = lib
    return lib


# Optional per-launch timing (bench.py's roofline leg): when KERNEL_TIMES is a dict, every launch made
# through `launch()` is bracketed by HIP events on the launch stream; `collect_kernel_times()` resolves them.
KERNEL_TIMES = None
_PENDING = []
PROF_BUFFER = None   # diagnostic: uint64[8] device tensor receiving in-kernel cycle sums of the MLP machine


def launch(name: str, *args, tag: str = ""):
    """Call one C-ABI entry point on the current stream and check its status."""
    fn = getattr(load(), name)
    if KERNEL_TIMES is None:
        check(fn(*args), name + (f"[{tag}]" if tag else ""))
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(*args)
    e1.record()
    _PENDING.append((name + (f"[{tag}]" if tag else ""), e0, e1))
    check(rc, name)


# Kernel timeline (include/hn_kernels.h, HnMlpArgs.timeline): when TIMELINE is a dict, every launch of the three machine
# kernels is handed a device uint64[8] slot per launch name and times ITSELF — also inside a HIP-graph replay, where no
# host-side event can sit between two kernels.  Enable BEFORE the step is captured (the pointers are baked into the
# graph); `timeline_reset()` zeroes the sums (e.g. after the warm-up), `timeline_read()` returns ms per name.
TIMELINE = None
TIMELINE_TICK_S = 1e-8        # wall_clock64(): 100 MHz


def timeline_slot(name: str, device) -> int:
    if TIMELINE is None:
        return 0
    t = TIMELINE.get(name)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise HnError(f"kernel timeline: first launch of {name} inside a stream capture (run a warm-up step first)")
        t = TIMELINE[name] = torch.zeros(8, dtype=torch.int64, device=device)
    return t.data_ptr()


def timeline_reset():
    torch.cuda.synchronize()
    for t in (TIMELINE or {}).values():
        t.zero_()
    torch.cuda.synchronize()


def timeline_read() -> dict:
    """{launch name: {'ms': summed duration, 'runs': launches, 'span_ms': first start .. last end}} (synchronises)."""
    torch.cuda.synchronize()
    out = {}
    for name, t in (TIMELINE or {}).items():
        v = t.cpu().tolist()
        out[name] = {"ms": v[4] * TIMELINE_TICK_S * 1e3, "runs": int(v[5]), "span_ms": (v[7] - v[6]) * TIMELINE_TICK_S * 1e3,
                     "first_start_tick": v[6], "last_end_tick": v[7]}
    return out


def collect_kernel_times():
    """ms per launch, grouped by name: {name: [ms, ...]} (synchronises)."""
    torch.cuda.synchronize()
    for name, e0, e1 in _PENDING:
        KERNEL_TIMES.setdefault(name, []).append(e0.elapsed_time(e1))
    _PENDING.clear()
    return KERNEL_TIMES


def check(rc: int, what: str):
    if rc != 0:
        kind = "argument error" if rc < 0 else "hipError_t"
        raise HnError(f"{what} failed: {kind} {rc}")


def stream_handle() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HnError("hypernerf_torch_amd runs on MI355X only: got a CPU tensor and there is no CPU fallback "
                          "(the CPU oracle lives in oracle/ and is test infrastructure)")


def ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def to_device_bytes(arr: np.ndarray, device) -> torch.Tensor:
    raw = np.ascontiguousarray(arr).view(np.uint8).reshape(-1)
    if raw.size == 0:
        raw = np.zeros(16, dtype=np.uint8)
    t = torch.from_numpy(raw.copy()).to(device)
    # tables are uploaded once and then read by launches on WHATEVER stream is current later on: make the upload
    # complete now instead of merely stream-ordered (never inside a stream capture, where the first use has long
    # happened in the warm-up runs)
    if t.is_cuda and not torch.cuda.is_current_stream_capturing():
        torch.cuda.current_stream(t.device).synchronize()
    return t
