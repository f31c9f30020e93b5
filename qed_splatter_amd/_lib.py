"""ctypes binding of libqed_splat.so (the C ABI declared in include/qed_splat.h).

This is the binding a maintainer of the reference would add instead of ``from gsplat.rendering
import rasterization`` (model.py:6-9).  There is deliberately NO fallback: if the library is
missing or a call fails, an exception is raised -- the product path never routes through the CPU
oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import types
from pathlib import Path

_LIB = None
LIB_PATH = Path(__file__).resolve().parent / "lib" / "libqed_splat.so"

# name -> (restype, argtypes); must list EVERY symbol include/qed_splat.h declares
_P = C.c_void_p
_I = C.c_int32
_L = C.c_int64
_F = C.c_float
_U = C.c_uint32
_D = C.c_double
SIGNATURES = {
    "qed_version": (C.c_int, []),
    "qed_last_error": (C.c_char_p, []),
    "qed_camera_setup": (C.c_int, [_I, _P, _P, _P, _P, _P]),
    "qed_project_fwd": (C.c_int, [_I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _P, _P, _I, _I, _I, _I, _F, _F, _F,
                                  _F, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "qed_project_bwd": (C.c_int, [_I, _I, _P, _P, _P, _P, _P, _I, _P, _I, _I, _P, _P, _I, _I, _F, _U, _P, _P, _P,
                                  _P, _P, _P, _P, _I, _P, _I, _P, _P, _P]),
    "qed_isect_scan": (C.c_int, [_P, _I, _P, _P, _L, _P, _P]),
    "qed_isect_emit": (C.c_int, [_I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _P, _L, _P, _P, _P]),
    "qed_sort_workspace_bytes": (_L, [_L]),
    "qed_sort_pairs": (C.c_int, [_P, _P, _P, _P, _P, _L, _I, _P, _L, _P, _P]),
    "qed_tile_offsets": (C.c_int, [_P, _P, _L, _I, _I, _I, _P, _P]),
    "qed_bin_workspace_bytes": (_L, [_L, _L]),
    "qed_bin_tiles": (C.c_int, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _L, _I, _P, _P, _P, _P, _P, _L, _P, _P, _P]),
    "qed_composite_fwd": (C.c_int, [_I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "qed_composite_bwd": (C.c_int, [_I, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "qed_loss_reduce": (C.c_int, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "qed_loss_grad": (C.c_int, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _P, _P, _P, _I, _F, _F, _P]),
    "qed_sh_grad_from_views": (C.c_int, [_I, _I, _P, _P, _L, _P, _L, _I, _F, _P, _I, _P, _I, _P]),
    "qed_pack_color_grad": (C.c_int, [_I, _P, _P, _P, _P]),
    "qed_host_device_pointer": (C.c_int, [_P, _P]),
    "qed_lr_exp_decay_dev": (C.c_int, [_P, _P, _F, _F, _I, _P]),
    "qed_densify_accumulate": (C.c_int, [_I, _P, _I, _P, _F, _P, _P, _P, _P]),
    "qed_densify_pos_ints": (C.c_int64, [_I]),
    "qed_densify_classify": (C.c_int, [_I, _P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _F, _F, _F, _P, _P, _P, _P]),
    "qed_densify_emit": (C.c_int, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "qed_densify_reset_opacity": (C.c_int, [_I, _P, _P, _P, _F, _P]),
    "qed_backproject_workspace_ints": (C.c_int64, [_I, _I, _I]),
    "qed_backproject_depth": (C.c_int, [_I, _I, _P, _F, _F, _F, _F, _P, _F, _I, _L, _P, _P, _P, _P, _P]),
    "qed_image_metrics": (C.c_int, [_I, _P, _P, _P, _P, _F, _P, _P, _P]),
    "qed_nanmean_exp": (C.c_int, [_I, _P, _I, _P, _P, _P]),
    "qed_step_metrics": (C.c_int, [_I, _P, _P, _P, _P, _F, _P, _I, _F, _P, _I, _I, _P, _F, _F, _F, _P, _P, _P, _P, _P]),
    "qed_ssim_maps_floats": (C.c_int64, [_I, _I]),
    "qed_ssim_sum_floats": (C.c_int64, [_I, _I]),
    "qed_ssim_fwd": (C.c_int, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "qed_ssim_fwd_step": (C.c_int, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _P]),
    "qed_ssim_bwd": (C.c_int, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P]),
    "qed_post_process_fwd": (C.c_int, [_I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "qed_post_process_bwd": (C.c_int, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "qed_image_losses_fwd": (C.c_int, [_I, _P, _P, _P, _P, _P, _F, _F, _P, _I, _F, _F, _P, _P, _P]),
    "qed_image_losses_bwd": (C.c_int, [_I, _P, _P, _P, _P, _P, _P, _F, _F, _P, _P, _I, _P, _P, _P]),
    "qed_loss_grad_ssim": (C.c_int, [_I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _P, _P, _P, _P, _I, _F, _P, _L,
                                     _P, _P]),
    "qed_image_losses_ssim_bwd": (C.c_int, [_I, _I, _P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _P, _P, _P, _P, _P, _L, _P]),
    "qed_adam_step": (C.c_int, [_P, _P, _P, _P, _I, _P, _P, _D, _D, _F, _I, _P, _P]),
    "qed_adam_step_dev": (C.c_int, [_P, _P, _P, _P, _I, _P, _P, _D, _D, _F, _P, _P, _P]),
    "qed_adam_step_sh": (C.c_int, [_P, _P, _P, _P, _I, _P, _P, _P, _D, _D, _F, _I, _P, _I, _F, _F, _I, _I, _I, _P, _I,
                                   _P, _L, _P, _L, _F, _I, _P, _P]),
}

class Post(C.Structure):
    """qed_post_t (include/qed_splat.h): get_outputs' post-processing outputs of qed_composite_fwd."""
    _fields_ = [("background", C.c_void_p), ("rgb", C.c_void_p), ("depth", C.c_void_p), ("tile_dmax", C.c_void_p)]


class PostGrad(C.Structure):
    """qed_post_grad_t: the gradients of those outputs, consumed by qed_composite_bwd."""
    _fields_ = [("background", C.c_void_p), ("render", C.c_void_p), ("v_rgb", C.c_void_p), ("v_depth", C.c_void_p)]


class AdamTick(C.Structure):
    """qed_adam_tick_t (include/qed_splat.h): the optimiser's device step state, advanced by qed_loss_grad_ssim."""
    _fields_ = [("dev_state", C.c_void_p), ("beta1", C.c_float), ("beta2", C.c_float), ("dev_lr_slot", C.c_void_p),
                ("lr_init", C.c_float), ("lr_final", C.c_float), ("max_steps", C.c_int32), ("skip_flag", C.c_void_p)]


# flags (include/qed_splat.h)
LOSS_SUMS_FLOATS = 8 + 4 * 1024          # QED_LOSS_SUMS_FLOATS
METRICS_WS_DOUBLES = 10 * 1024           # QED_METRICS_WS_DOUBLES
STEP_METRICS_WS_DOUBLES = 16 * 1024      # QED_STEP_METRICS_WS_DOUBLES
F_ANTIALIASED = 1
F_LOG_SCALES = 2
F_LOGIT_OPAC = 4
F_DEPTH_CHANNEL = 8
F_SIGMOID_COLORS = 16
F_TIGHT_TILES = 32
F_CAMERA_C2W = 128
F_SH_GRAD_COMPACT = 64
SPLAT_FLOATS = 12
VSPLAT_FLOATS = 16
SH_JAC_FLOATS = 10            # QED_SH_JAC_FLOATS
STATUS_WORDS = 4
TILE = 16
CL_TILE_WAVES, CL_QUADRANT_WAVES, CL_HALF_AND_HALF, CL_NO_CULL, CL_ORDER_READY = 1, 2, 3, 4, 8
BIN_AUTO, BIN_TWO_STAGE, BIN_TILE_SORT = 0, 1, 2


def bin_mode() -> int:
    """Test / measurement hook (auto in production): QED_BIN_MODE=two_stage|tile_sort forces one binning pipeline."""
    return {"two": BIN_TWO_STAGE, "til": BIN_TILE_SORT}.get(os.environ.get("QED_BIN_MODE", "")[:3], BIN_AUTO)


def composite_launch_flags() -> int:
    """Test hooks (0 in production): QED_COMPOSITE_WAVES=tile|quadrant|half forces one launch shape of the
    compositing kernels, QED_COMPOSITE_NOCULL=1 turns their quadrant culling off.  Read HERE, by the host layer; the
    library itself reads no environment."""
    f = {"t": CL_TILE_WAVES, "q": CL_QUADRANT_WAVES, "h": CL_HALF_AND_HALF}.get(
        os.environ.get("QED_COMPOSITE_WAVES", "")[:1], 0)
    if os.environ.get("QED_COMPOSITE_NOCULL", "")[:1] == "1":
        f |= CL_NO_CULL
    return f


class QedSplatError(RuntimeError):
    pass


ABI_VERSION = 3          # include/qed_splat.h: QED_ABI_VERSION (tests/test_abi.py checks the two against each other)


def load():
    """Load the shared library (once).  Raises if it has not been built, or implements another version of the C ABI."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch first: it ships its own libamdhip64; loading ours after it makes the dynamic loader
    # resolve both to the ONE runtime torch's streams and allocations belong to
    import torch  # noqa: F401
    path = Path(os.environ.get("QED_SPLAT_LIB", LIB_PATH))
    if not path.exists():
        raise QedSplatError(
            f"{path} not found: build it with `python -m qed_splatter_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback on the product path.")
    cdll = C.CDLL(str(path))
    ns = {"_cdll": cdll}
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(cdll, name)         # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
        ns[name] = _wrap(name, fn) if res is C.c_int and args else fn
    got = int(cdll.qed_version())
    if got != ABI_VERSION:
        raise QedSplatError(f"{path} implements C ABI version {got}, this package binds version {ABI_VERSION} "
                            "(include/qed_splat.h: QED_ABI_VERSION): rebuild it with `python -m qed_splatter_amd.build --force`")
    _LIB = types.SimpleNamespace(**ns)
    return _LIB


def _wrap(name, fn):
    def call(*a):
        if TIMER.active:
            tok = TIMER.begin(name)
            rc = fn(*a)
            TIMER.end(tok)
            return rc
        if MARKERS:                          # QED_ROCTX=1: a roctx range per entry point (rocprofv3 --marker-trace)
            push, pop = MARKERS
            push(name)
            try:
                return fn(*a)
            finally:
                pop()
        return fn(*a)
    call.__name__ = name
    return call


def _roctx_markers():
    """roctx ranges around the entry points K1-K8 are reached through, named after them, for
    `rocprofv3 --marker-trace --kernel-trace`.  Off unless QED_ROCTX=1 (two extra host calls per entry point); through
    torch.cuda.nvtx, which is roctx on a ROCm build of PyTorch."""
    if os.environ.get("QED_ROCTX", "") != "1":
        return None
    try:
        import torch
        nv = torch.cuda.nvtx
        nv.range_push("qed_splat")
        nv.range_pop()
        return (nv.range_push, nv.range_pop)
    except Exception:                        # (a torch without the marker library: tracing is an aid, not a requirement)
        return None


MARKERS = _roctx_markers()


class KernelTimer:
    """Optional per-entry-point HIP-event timing on the stream the kernels are launched on
    (bench.py's roofline leg).  Inactive (zero overhead beyond one attribute test) by default."""

    def __init__(self):
        self.active = False
        self.events = {}

    def begin(self, name: str):
        import torch
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return (name, ev)

    def end(self, tok) -> None:
        import torch
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.events.setdefault(tok[0], []).append((tok[1], ev))

    def summary(self):
        """name -> (calls, mean ms); call after torch.cuda.synchronize()."""
        out = {}
        for name, pairs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            out[name] = (len(ms), sum(ms) / max(len(ms), 1))
        return out

    def reset(self):
        self.events = {}


TIMER = KernelTimer()


def current_stream() -> int:
    """Raw handle of the current stream of the current device (what every entry point takes as ``stream``).
    ``torch.cuda.current_stream().cuda_stream`` costs ~10 us of Python per call -- a step of the reference-shaped route
    asks ten times -- so the two C accessors behind it are called directly where this torch has them."""
    import torch
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:                                  # (another torch: the public, slower spelling)
        return torch.cuda.current_stream().cuda_stream


def check(rc: int, what: str) -> int:
    if rc < 0:
        msg = load().qed_last_error().decode("utf-8", "replace")
        raise QedSplatError(f"{what} failed (rc={rc}): {msg}")
    return rc


def ptr(t) -> int:
    """Device pointer of a tensor (0 for None)."""
    return 0 if t is None else t.data_ptr()
