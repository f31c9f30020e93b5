"""Drop-in for ``gsplat.rendering.rasterization`` as the reference calls it
(/root/reference/qed_splatter/model.py:267-288): same keyword names, same return triple
``(render[C,H,W,3|4], alpha[C,H,W,1], info)`` and the ``info`` keys the reference and Nerfstudio's
densification strategy read (model.py:289-292).

Host code is Python + PyTorch-ROCm (device memory, streams, autograd plumbing); every arithmetic
step runs in hand-written gfx950 HIP kernels behind the C ABI of ``include/qed_splat.h``.

Autograd graph (mirrors gsplat's, so ``info["means2d"].retain_grad()`` at model.py:289-290 works):

    means, quats, scales, opacities, colors, viewmats
        |  _ProjectSH   (qed_project_fwd / qed_project_bwd: projection + SH colour, fused)
    means2d, depths, conics, opac, rgb      <- info["means2d"] is this non-leaf tensor
        |  _Composite   (isect scan/emit -> radix sort -> tile offsets -> qed_composite_fwd/bwd)
    render, alpha

``_Composite.backward`` returns strided views of ONE packed 64-byte-per-Gaussian gradient buffer;
``_ProjectSH.backward`` recognises them and hands the packed buffer to the fused kernel, so no
gradient is ever repacked on the hot path.
"""
from __future__ import annotations

import math
import time
import warnings
import weakref
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L


_stream = L.current_stream


def _f32c(t: Tensor, name: str) -> Tensor:
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    if not t.is_cuda:
        raise L.QedSplatError(f"{name} must live on the GPU: this operator has no CPU path")
    return t.contiguous()


def tile_bits_for(n_tiles: int) -> int:
    return int(math.floor(math.log2(n_tiles))) + 1


class _Workspace:
    """Per-device state that persists across calls: the status words and the intersection capacity.

    Key/value buffers are taken from PyTorch's caching allocator per call (sized by a capacity
    that only grows) and stay alive until that call's backward has run.  The kernels read the
    actual count M from device memory, so launches never need M on the host.

    The capacity is calibrated per SHAPE ``(width, height, N, C)``: the first call of a shape reads M back (one host
    read) whatever the caller asked for -- the reference's resolution schedule quadruples the pixel count at steps 3000
    and 6000 (model.py:244-250) and every densification changes N (config.py:40-41), and either can multiply M.  Calls
    of a calibrated shape may run without the read-back (``_sync=False``); their M still comes back, one call late
    (``poll_pending``), and keeps the capacity at ``HEADROOM`` x the longest list seen.  Should a frame overflow all
    the same, it renders empty, the status word makes the optimiser launches enqueued behind it no-ops
    (``qed_adam_step*``'s ``skip_flag``), and the next call regrows the buffer, warns and reads M back again: no
    exception a step late, no update from an empty frame.
    """

    HEADROOM = 2.0          # capacity / longest list seen: grids are sized by it but only M entries are ever touched
    POLL_TIMEOUT_S = 20.0   # how long poll_pending waits for a frame's count before it asks the device what happened

    def __init__(self, device):
        self.device = device
        self.capacity = 0
        # [M of the last binning call | status words]: adjacent, so that an asynchronous call reads them back in ONE copy
        self.words = torch.zeros(1 + L.STATUS_WORDS, dtype=torch.int32, device=device)
        self.n_isect = self.words[:1]
        self.status = self.words[1:]
        self._ring, self._ring_at = [], 0     # pinned read-back buffers of the asynchronous calls, reused
        self.pending = None          # (pinned host words [M, overflow, watchdog, 0], shape key, frame) of an async call
        self.frame = 0               # counts the asynchronous frames armed on this device (arm_pending)
        self.last_overflow = False   # an asynchronous frame overflowed and nobody has been told yet (take_overflow_flag)
        self.m_seen: Dict[tuple, int] = {}   # shape key -> longest list read back for that shape
        self.force_sync = False      # the next call reads M back (an asynchronous frame overflowed)
        self.overflows = 0           # asynchronous frames that rendered empty (diagnostics / tests)
        # optimisers that keep a HOST step counter and take this workspace's overflow word as skip_flag: told when a step
        # was skipped on the device, so that their bias corrections stay in step with their moments
        self.steppers: "weakref.WeakSet" = weakref.WeakSet()
        self.waited_s = 0.0          # host time spent in poll_pending waiting for the device to reach the previous frame's binning
        self.host_words_ok = True    # False once a count failed to arrive through pinned memory: read-backs by copy from then on

    def host_slot(self):
        """A pinned 4-word buffer for one asynchronous read-back, as (numpy view, address): qed_bin_tiles' last list kernel
        stores {M, overflow, watchdog, 0} into it (``host_words``: pinned memory is mapped into the device's address space)
        and ``poll_pending`` watches word 0 turn from -1 into M.  Neither a copy nor an event enters the stream: the 12-byte
        device-to-host copy and the event that used to sit behind the binning cost ~10 us between it and the compositing
        pass (a blit kernel and two barrier packets).  At most one read-back is pending at a time (the next call polls it
        before it arms its own), so two slots used in turn are never overwritten in flight."""
        if not self._ring:
            import ctypes
            lib = L.load()
            for _ in range(2):
                host = torch.zeros(4, dtype=torch.int32).pin_memory()
                # the address the KERNEL stores to: the runtime's device-side alias of the pinned block (the same number
                # under unified addressing, but not for memory pinned by registration)
                dptr = ctypes.c_void_p()
                L.check(lib.qed_host_device_pointer(host.data_ptr(), ctypes.addressof(dptr)), "qed_host_device_pointer")
                self._ring.append((host.numpy(), dptr.value, host))          # (the tensor keeps the memory alive)
        self._ring_at ^= 1
        words, ptr, _keep = self._ring[self._ring_at]
        words[0] = -1
        return words, ptr

    def reset(self) -> None:
        """Forget every calibration (the next call of any shape reads M back and sizes the buffer afresh)."""
        self.poll_pending()
        self.capacity, self.m_seen, self.force_sync = 0, {}, False

    def calibrated(self, key) -> bool:
        return self.capacity > 0 and key in self.m_seen and not self.force_sync

    def saw(self, key, M: int) -> None:
        self.m_seen[key] = max(self.m_seen.get(key, 0), int(M))
        if len(self.m_seen) > 64:                                   # (cameras of many sizes: keep the table small)
            self.m_seen.pop(next(iter(self.m_seen)))
        self.capacity = max(self.capacity, int(M * self.HEADROOM) + 4096)

    def skip_flag_ptr(self) -> int:
        """Device address of the overflow word: what the optimiser launches take as ``skip_flag``."""
        return self.status.data_ptr()

    def arm_pending(self, words, key) -> None:
        """An asynchronous frame has been enqueued whose {M, overflow} will land in ``words``: the next call looks."""
        self.frame += 1
        self.pending = (words, key, self.frame)

    def counted_step(self, opt) -> None:
        """An optimiser with a host step counter has counted a step whose launch sits behind the current frame (and takes
        its overflow word as skip_flag): if that frame turns out to have overflowed, THAT optimiser takes the step back --
        not every optimiser of the device (another model's, one that did not step this iteration)."""
        opt.__dict__["_qed_frame"] = self.frame

    def take_overflow_flag(self) -> bool:
        """Did an asynchronous frame overflow since this was last asked?  (get_outputs puts it into ``info``.)"""
        flag, self.last_overflow = self.last_overflow, False
        return flag

    def poll_pending(self) -> None:
        if self.pending is None:
            return
        words, key, frame = self.pending
        self.pending = None
        if words[0] < 0:
            # the host is a frame ahead of the device: wait for that frame's binning.  A few looks back to back (the word
            # usually lands within microseconds), then sleeps that double up to 100 us: a host that runs ahead of a ~1 ms
            # step does not burn a core on it, and wakes at most a tenth of a step late
            t_wait = time.monotonic()
            deadline = t_wait + self.POLL_TIMEOUT_S
            looks, nap = 0, 5e-6
            while words[0] < 0:
                looks += 1
                if looks <= 8:
                    time.sleep(0)
                    continue
                time.sleep(nap)
                nap = min(2.0 * nap, 1e-4)
                if time.monotonic() > deadline:
                    torch.cuda.synchronize(self.device)             # surfaces a device fault as its own error
                    if words[0] < 0:
                        # the frame has finished and its words are not here: the store into pinned memory does not reach
                        # this host (a runtime / allocator configuration this was not tested on).  Take the frame's words
                        # from device memory -- nothing has run since -- and read every later count back by copy.
                        dev_words = self.words[:3].tolist()
                        words[0], words[1], words[2] = dev_words[0], dev_words[1], dev_words[2]
                        self.host_words_ok = False
                        warnings.warn("qed_splatter_amd: the intersection count did not arrive through pinned host memory; "
                                      "falling back to a synchronous read-back per call", RuntimeWarning, stacklevel=3)
            self.waited_s += time.monotonic() - t_wait
        M, overflow, watchdog = int(words[0]), int(words[1]), int(words[2])
        if watchdog:
            self.status.zero_()
            raise L.QedSplatError("the radix-sort look-back watchdog fired in the previous asynchronous rasterization: "
                                  "that frame's list was mis-sorted")
        if overflow:
            # that frame rendered empty and the optimiser launches behind it did nothing (skip_flag); clear the word
            # (stream-ordered behind those launches), make room and read M back on the call that follows
            self.status.zero_()
            self.overflows += 1
            self.force_sync = True
            old = self.capacity
            self.saw(key, overflow)
            self.last_overflow = True
            for opt in list(self.steppers):                        # (the host is at most one frame ahead: ONE step was skipped)
                if opt.__dict__.get("_qed_frame") == frame:        # ... by the optimisers that stepped behind THAT frame
                    opt.__dict__["_qed_frame"] = None
                    opt.on_skipped_step()
            warnings.warn(f"qed_splatter_amd: an asynchronous rasterization needed {overflow} tile intersections, more "
                          f"than the buffer held ({old}); that frame rendered empty and its optimiser step was skipped "
                          f"on the device.  Capacity raised to {self.capacity}.", RuntimeWarning, stacklevel=3)
            return
        self.saw(key, M)


_WORKSPACES: Dict[int, _Workspace] = {}

# qed_project_fwd's tile_masks: with the tight rectangles (QED_F_TIGHT_TILES) list only the tiles some pixel of which can reach
# alpha >= 1/255 (tests switch it off to compare the lists; the images must not change)
EXACT_TILE_LISTS = True
_NO_MASKS: "Dict[torch.device, Tensor]" = {}

# qed_project_fwd's sh_jac hand-over to qed_project_bwd (tests switch it off to keep the coefficient-reading backward
# kernels, which direct C-ABI callers without the planes still get, under the same parity checks)
SH_HANDOVER = True

# qed_composite_fwd's t_final image for qed_composite_bwd (tests switch it off to keep gsplat's T_final = 1 - alpha
# under the same parity checks; see tests/test_gpu_parity.py::test_dense_scene_gradients_against_the_fp32_band)
KEEP_T_FINAL = True

# Called by _ProjectSH.backward right before qed_project_bwd is launched, with (packed gradient rows [C*N,16], sh_jac planes,
# C * N), when the backward pass writes compact SH gradients: the data-parallel exchange packs the colour-gradient message
# from them and puts its all-gather on the links while the projection backward runs (parallel.early_gather).  Keyed by the
# storage of the model's `means` (the flat parameter buffer): a hook belongs to ONE model, and two models (or threads) running
# their backward passes at once never see each other's.
PRE_PROJECT_BWD: "Dict[int, object]" = {}


def _workspace(device) -> _Workspace:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    ws = _WORKSPACES.get(idx)
    if ws is None:
        ws = _WORKSPACES[idx] = _Workspace(torch.device("cuda", idx))
    return ws


# ==================================================================================================
# the two nodes driven by hand (captured segments: segments.py)
# ==================================================================================================
class ManualContext:
    """Stands in for the autograd context when _ProjectSH / _Composite are run WITHOUT the autograd engine: a captured
    get_outputs segment calls their forward and, in a second hipGraph, their backward directly.  (The engine pins gradient
    accumulation to the stream a leaf first ran on and synchronises with it at the end of every backward pass: on the
    legacy default stream -- where Nerfstudio trains -- that is an illegal call inside a stream capture.)"""

    def __init__(self, needs_input_grad):
        self.needs_input_grad = tuple(needs_input_grad)
        self.saved_tensors = ()

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors

    def set_materialize_grads(self, _value):
        pass

    def mark_non_differentiable(self, *_tensors):
        pass


def _run_node(fn, manual, needs, *args):
    """fn.apply(*args), or -- driven by hand -- fn.forward on a ManualContext that is kept in ``manual`` for
    manual_backward().  ``needs``: needs_input_grad of the hand-driven form (None: from the arguments)."""
    if manual is None:
        return fn.apply(*args)
    if needs is None:
        needs = [torch.is_tensor(a) and a.requires_grad for a in args]
    ctx = ManualContext(needs)
    manual.append(ctx)
    with torch.no_grad():
        return fn.forward(ctx, *args)


def manual_backward(manual, v_render=None, v_alpha=None, v_rgb=None, v_depth=None):
    """The backward pass of a rasterization(_manual=[...]) call: (v_means, v_quats, v_scales, v_opacities, v_colors | v_sh0,
    v_sh_rest, v_viewmats) for upstream gradients of render / alpha / post_rgb / post_depth (any may be None)."""
    c_proj, c_comp = manual
    with torch.no_grad():
        g = _Composite.backward(c_comp, v_render, v_alpha, None, v_rgb, v_depth)
        v_means2d, v_conics, v_rgb_g, v_opac, v_depths = g[:5]
        return _ProjectSH.backward(c_proj, v_means2d, v_depths, v_conics, v_opac, v_rgb_g)[:7]


# ==================================================================================================
# projection + SH
# ==================================================================================================
class _ProjectSH(torch.autograd.Function):
    # what forward saves, by NAME: code outside this class (segments.py, parallel.py) asks saved(ctx, "sh_jac") instead of
    # counting positions -- a reordering of the two statements in forward only has to keep these two tuples in step
    SAVED = ("means", "quats", "scales", "opacities", "sh0", "shN", "viewmats", "Ks", "radii", "sh_jac")
    META = ("N", "C", "width", "height", "sh_degree", "flags", "eps2d", "sh0_stride", "shN_stride")

    @classmethod
    def saved(cls, ctx, name: str):
        """A saved tensor or meta value of a forward context / backward node of this Function, by name."""
        if name in cls.SAVED:
            t = ctx.saved_tensors
            assert len(t) == len(cls.SAVED)
            return t[cls.SAVED.index(name)]
        assert len(ctx.meta) == len(cls.META)
        return ctx.meta[cls.META.index(name)]

    @staticmethod
    def forward(ctx, means, quats, scales, opacities, sh0, shN, viewmats, Ks, width, height, tile_w, tile_h,
                sh_degree, flags, eps2d, near_plane, far_plane, radius_clip, c2w=None, intr=None, lazy_sh=None):
        lib = L.load()
        ctx.lazy_sh = lazy_sh
        # undefined output gradients arrive as None instead of freshly zero-filled tensors (autograd would
        # otherwise fill one per output per step, the 24 MB splat record included)
        ctx.set_materialize_grads(False)
        N, C = means.shape[0], viewmats.shape[0]
        dev = means.device
        means, quats, scales = _f32c(means, "means"), _f32c(quats, "quats"), _f32c(scales, "scales")
        opac_shape = opacities.shape
        opacities = _f32c(opacities, "opacities").reshape(-1)
        sh0 = _f32c(sh0, "colors")
        shN = _f32c(shN, "colors") if shN is not None else None
        viewmats, Ks = _f32c(viewmats, "viewmats"), _f32c(Ks, "Ks")
        cam_in = (viewmats, Ks)
        if c2w is not None:
            # QED_F_CAMERA_C2W: the kernel derives the view matrices from the camera-to-world matrices itself and fills
            # `viewmats` / `Ks` (the caller's uninitialised buffers) for the backward pass and everything downstream
            assert c2w.dtype == torch.float32 and c2w.is_contiguous() and tuple(c2w.shape) == (C, 3, 4)
            assert intr.dtype == torch.float32 and intr.is_contiguous() and tuple(intr.shape) == (C, 4)
            assert not viewmats.requires_grad, "no gradient reaches camera-to-world matrices through this path"
            cam_in, flags = (c2w, intr), flags | L.F_CAMERA_C2W
        sh0_stride = sh0.stride(0) if sh0.dim() > 1 else 3
        # a [N,K,3] colours tensor is passed as (colors, 3K, colors + 3, 3K) -- no copy
        if sh0.dim() == 3:
            K = sh0.shape[1]
            sh0_flat = sh0
            shN_ptr = sh0.data_ptr() + 12 if K > 1 else 0
            shN_stride = 3 * K
            sh0_stride = 3 * K
        else:
            sh0_flat = sh0
            shN_ptr = L.ptr(shN)
            shN_stride = shN.stride(0) if shN is not None else 0

        radii = torch.empty(C, N, dtype=torch.int32, device=dev)
        means2d = torch.empty(C, N, 2, dtype=torch.float32, device=dev)
        depths = torch.empty(C, N, dtype=torch.float32, device=dev)
        conics = torch.empty(C, N, 3, dtype=torch.float32, device=dev)
        opac = torch.empty(C, N, dtype=torch.float32, device=dev)
        rgb = torch.empty(C, N, 3, dtype=torch.float32, device=dev)
        splats = torch.empty(C * N, L.SPLAT_FLOATS, dtype=torch.float32, device=dev)
        tiles_per_gauss = torch.empty(C, N, dtype=torch.int32, device=dev)
        # exact tile lists (with the tight rectangles): which tiles of a Gaussian's rectangle any of its pixels can reach
        tile_masks = torch.empty(C * N, 2, dtype=torch.int64, device=dev) if (flags & L.F_TIGHT_TILES) and EXACT_TILE_LISTS else None
        n_blocks = (C * N + 255) // 256
        block_sums = torch.empty(max(n_blocks, 1), dtype=torch.int32, device=dev)
        # what the backward pass needs of the SH part (direction Jacobian + clamp mask, 40 B per slot) instead of
        # re-reading the 3 K coefficients and rebuilding the basis derivatives: qed_project_fwd's sh_jac
        sh_jac = None
        if SH_HANDOVER and sh_degree >= 0 and any(ctx.needs_input_grad[:7]):
            sh_jac = torch.empty(L.SH_JAC_FLOATS, C * N, dtype=torch.float32, device=dev)
        L.check(lib.qed_project_fwd(
            N, C, L.ptr(means), L.ptr(quats), L.ptr(scales), L.ptr(opacities), L.ptr(sh0_flat), sh0_stride,
            shN_ptr, shN_stride, sh_degree, L.ptr(cam_in[0]), L.ptr(cam_in[1]), width, height, tile_w, tile_h, eps2d,
            near_plane, far_plane, radius_clip, flags, L.ptr(radii), L.ptr(means2d), L.ptr(depths), L.ptr(conics),
            L.ptr(opac), L.ptr(rgb), L.ptr(splats), L.ptr(tiles_per_gauss), L.ptr(tile_masks), L.ptr(block_sums),
            L.ptr(viewmats) if c2w is not None else None, L.ptr(Ks) if c2w is not None else None, L.ptr(sh_jac),
            _stream()), "qed_project_fwd")
        flags &= ~L.F_CAMERA_C2W                  # (the backward pass reads the view matrices the kernel wrote)
        ctx.save_for_backward(means, quats, scales, opacities, sh0, shN, viewmats, Ks, radii, sh_jac)      # order: SAVED
        ctx.meta = (N, C, width, height, sh_degree, flags, eps2d, sh0_stride, shN_stride)                  # order: META
        ctx.opac_shape = opac_shape
        if tile_masks is None:
            tile_masks = _NO_MASKS.get(dev)
            if tile_masks is None:
                tile_masks = _NO_MASKS[dev] = torch.empty(0, dtype=torch.int64, device=dev)
        ctx.mark_non_differentiable(radii, splats, tiles_per_gauss, block_sums, tile_masks)
        return means2d, depths, conics, opac, rgb, radii, splats, tiles_per_gauss, block_sums, tile_masks

    @staticmethod
    def backward(ctx, v_means2d, v_depths, v_conics, v_opac, v_rgb, *_unused):
        lib = L.load()
        means, quats, scales, opacities, sh0, shN, viewmats, Ks, radii, sh_jac = ctx.saved_tensors
        N, C, width, height, sh_degree, flags, eps2d, sh0_stride, shN_stride = ctx.meta
        dev = means.device
        vsplat = _packed_vsplat(C, N, v_means2d, v_depths, v_conics, v_opac, v_rgb, dev)

        # All parameter gradients live in ONE allocation, in the order
        # (means, scales, quats, opacities, sh0, shN) = qed_splatter_amd.model.GROUP_ORDER, so a
        # caller can all-reduce / Adam-step them as a single contiguous range.
        k_active = (sh_degree + 1) ** 2 if sh_degree >= 0 else 1
        n_sh0 = sh0.numel()
        n_shN = shN.numel() if shN is not None else 0
        flat = torch.empty(11 * N + n_sh0 + n_shN, dtype=torch.float32, device=dev)
        v_means = flat[0:3 * N].view(N, 3)
        v_scales = flat[3 * N:6 * N].view(N, 3)
        v_quats = flat[6 * N:10 * N].view(N, 4)
        v_opacities = flat[10 * N:11 * N]
        v_sh0 = flat[11 * N:11 * N + n_sh0].view(sh0.shape)
        v_shN = flat[11 * N + n_sh0:].view(shN.shape) if shN is not None else None
        if sh0.dim() == 3:                      # colours [N,K,3] in one tensor
            if k_active < sh0.shape[1]:
                v_sh0.zero_()                   # coefficients above the active degree get zero gradient
            v_sh0_ptr, v_sh0_stride = v_sh0.data_ptr(), sh0_stride
            v_shN_ptr, v_shN_stride = v_sh0.data_ptr() + 12, shN_stride
            shN_ptr = sh0.data_ptr() + 12
        else:
            # (with F_SH_GRAD_COMPACT the kernel leaves v_shN untouched: qed_sh_grad_from_views fills the active
            # coefficients later, the inactive ones still need their zeros)
            if shN is not None and k_active - 1 < shN.shape[1]:
                v_shN.zero_()
            v_sh0_ptr, v_sh0_stride = v_sh0.data_ptr(), sh0_stride
            v_shN_ptr, v_shN_stride = L.ptr(v_shN), shN_stride
            shN_ptr = L.ptr(shN)
        del flat                                # only the views stay referenced -> autograd can adopt them
        v_viewmats = None
        if ctx.needs_input_grad[6]:
            v_viewmats = torch.zeros_like(viewmats)
        # QED_F_SH_GRAD_COMPACT asked for by a caller that keeps the Gaussians' SH gradients in compact form until somebody
        # reads them (model.QEDSplatterModel, lazy_sh_grad): it is asked NOW whether this backward pass may still write that
        # form -- not when gradients of an earlier pass are waiting to be added to
        lazy = getattr(ctx, "lazy_sh", None)
        if (flags & L.F_SH_GRAD_COMPACT) and lazy is not None and not lazy(v_sh0, v_shN, viewmats, sh_degree):
            flags &= ~L.F_SH_GRAD_COMPACT
        if PRE_PROJECT_BWD and sh_jac is not None and (flags & L.F_SH_GRAD_COMPACT):
            hook = PRE_PROJECT_BWD.get(means.untyped_storage().data_ptr())
            if hook is not None:
                hook(vsplat, sh_jac, C * N)
        L.check(lib.qed_project_bwd(
            N, C, L.ptr(means), L.ptr(quats), L.ptr(scales), L.ptr(opacities), L.ptr(sh0), sh0_stride, shN_ptr,
            shN_stride, sh_degree, L.ptr(viewmats), L.ptr(Ks), width, height, eps2d, flags, L.ptr(radii),
            L.ptr(vsplat), L.ptr(v_means), L.ptr(v_quats), L.ptr(v_scales), L.ptr(v_opacities), v_sh0_ptr,
            v_sh0_stride, v_shN_ptr, v_shN_stride, L.ptr(v_viewmats), L.ptr(sh_jac), _stream()), "qed_project_bwd")
        v_opacities = v_opacities.view(ctx.opac_shape)
        return (v_means, v_quats, v_scales, v_opacities, v_sh0, v_shN, v_viewmats, None) + (None,) * 13


_VSPLAT_REGISTRY: "weakref.WeakValueDictionary[int, Tensor]" = weakref.WeakValueDictionary()


def _packed_vsplat(C, N, v_means2d, v_depths, v_conics, v_opac, v_rgb, dev) -> Tensor:
    """Return the packed [C*N,16] gradient buffer.

    Fast path: the incoming gradients are exactly the strided views _Composite.backward produced
    of one packed buffer (same storage, expected offsets and strides) -> use that buffer as is.
    Otherwise (user-supplied or autograd-summed gradients) pack them."""
    R = L.VSPLAT_FLOATS
    base = None
    if v_means2d is not None:
        base = _VSPLAT_REGISTRY.get(v_means2d.untyped_storage().data_ptr())

    def is_view(g, col, width):
        if g is None or g.untyped_storage().data_ptr() != base.untyped_storage().data_ptr():
            return False
        want = (N * R, R, 1) if width > 1 else (N * R, R)
        shape = (C, N, width) if width > 1 else (C, N)
        return g.storage_offset() == col and tuple(g.shape) == shape and tuple(g.stride()) == want

    if base is not None and base.numel() == C * N * R and is_view(v_means2d, 0, 2) and is_view(v_conics, 4, 3) \
            and is_view(v_opac, 7, 1) and is_view(v_rgb, 8, 3) and (v_depths is None or is_view(v_depths, 11, 1)):
        return base
    out = torch.zeros(C * N, R, dtype=torch.float32, device=dev)
    if v_means2d is not None:
        out[:, 0:2] = v_means2d.reshape(C * N, 2)
    if v_conics is not None:
        out[:, 4:7] = v_conics.reshape(C * N, 3)
    if v_opac is not None:
        out[:, 7] = v_opac.reshape(C * N)
    if v_rgb is not None:
        out[:, 8:11] = v_rgb.reshape(C * N, 3)
    if v_depths is not None:
        out[:, 11] = v_depths.reshape(C * N)
    return out


# ==================================================================================================
# tile binning + sort + compositing
# ==================================================================================================
def _bin_and_sort(N, C, means2d, radii, depths, tiles_per_gauss, block_sums, tile_w, tile_h, sync=True, splats=None,
                  size=None, tile_masks=None, capture_slot=None):
    """Tile binning (qed_bin_tiles): one C call that leaves the list sorted by (camera, tile, depth) and the tile
    offsets.

    Returns (isect_ids, flatten_ids, offsets, M).  With ``sync=False`` -- honoured only for a shape
    ``(width, height, N, C)`` whose capacity has been calibrated by a synchronous call (_Workspace) -- nothing is read
    back: M and isect_ids are None and flatten_ids keeps its capacity length.
    ``capture_slot`` (while a hipGraph is being captured): the pinned (words, address) pair every REPLAY of the captured
    launch stores {M, overflow, watchdog} into -- the replaying code sets words[0] = -1 before a replay and hands the pair
    to ``_Workspace.pending`` after it (segments.OutputsSegment), exactly as an eager asynchronous call does.
    """
    lib = L.load()
    dev = means2d.device
    ws = _workspace(dev)
    n_tiles = tile_w * tile_h
    key = (tuple(size) if size is not None else (tile_w, tile_h), N, C)
    n_isect = ws.n_isect               # (read by the kernels of this call only; launches are stream-ordered)
    offsets = torch.empty(C * n_tiles + 1, dtype=torch.int32, device=dev)
    capturing = torch.cuda.is_current_stream_capturing()
    if capturing:
        # inside a hipGraph capture nothing may touch the host: capacity is frozen at its calibrated
        # value and the overflow word is polled by the replaying code (graph.GraphedTrainStep)
        if ws.capacity == 0 or key not in ws.m_seen:
            raise L.QedSplatError("run one eager call of this shape (image size, number of Gaussians, cameras) before "
                                  "capturing: it calibrates the intersection capacity")
        sync = False
    else:
        ws.poll_pending()
        if not ws.calibrated(key) or not ws.host_words_ok:
            sync = True                                   # first call of a shape, or the call after an overflow
    if ws.capacity == 0:
        ws.capacity = max(1 << 16, 8 * C * N)
    # which pipeline (both give the same list bit for bit): by the longest list this shape has produced, not by the
    # (generously padded) capacity the library's own QED_BIN_AUTO would go by
    mode = L.bin_mode()
    if mode == L.BIN_AUTO and ws.m_seen.get(key, 0) > 0:
        mode = L.BIN_TILE_SORT if int(1.25 * ws.m_seen[key]) <= 1024 * C * n_tiles else L.BIN_TWO_STAGE
    host = host_ptr = None
    if not sync and not capturing:
        host, host_ptr = ws.host_slot()
    elif capturing and capture_slot is not None:
        host, host_ptr = capture_slot
    for _attempt in range(2):
        cap = ws.capacity
        flatten_ids = torch.empty(cap, dtype=torch.int32, device=dev)
        isect_ids = torch.empty(cap, dtype=torch.int64, device=dev) if sync else None
        scratch = torch.empty(int(lib.qed_bin_workspace_bytes(C * N, cap)), dtype=torch.uint8, device=dev)
        L.check(lib.qed_bin_tiles(N, C, L.ptr(means2d), L.ptr(radii), L.ptr(depths), L.ptr(tiles_per_gauss), L.ptr(splats),
                                  L.ptr(tile_masks), L.ptr(block_sums), tile_w, tile_h, cap, mode, L.ptr(flatten_ids), L.ptr(offsets),
                                  L.ptr(n_isect), L.ptr(isect_ids), L.ptr(scratch), scratch.numel(), L.ptr(ws.status),
                                  host_ptr, _stream()), "qed_bin_tiles")
        if capturing:
            ws.last_n_isect = n_isect                  # device tensor the replaying code polls
            return None, flatten_ids, offsets, None
        if not sync:
            ws.arm_pending(host, key)
            return None, flatten_ids, offsets, None
        # one host read: M, the overflow word and the look-back watchdog word
        host = ws.words[:3].tolist()
        M, overflow = int(host[0]), int(host[1])
        if host[2]:
            ws.status.zero_()
            raise L.QedSplatError("the radix-sort look-back watchdog fired: the list of this frame is mis-sorted")
        if overflow == 0:
            ws.saw(key, M)
            ws.force_sync = False
            return isect_ids[:M], flatten_ids[:M], offsets, M
        ws.status.zero_()
        ws.saw(key, overflow)
    raise L.QedSplatError("intersection buffer overflow persisted after regrowth")


class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, conics, rgb, opac, depths, splats, flatten_ids, offsets, backgrounds, width, height,
                tile_w, tile_h, channels, absgrad, vsplat_holder=None, post_background=None, grad_leaf=None,
                tile_order=None):
        lib = L.load()
        ctx.set_materialize_grads(False)
        ctx.vsplat_holder = vsplat_holder
        ctx.grad_leaf = grad_leaf[0] if grad_leaf else None      # (in a list: not an autograd input)
        # tile_order: a launch order for THIS kernel from an earlier frame of the same camera -- a tensor, or a camera's
        # SLOT [tensor, valid] (model.py: _frame_orders): read while valid, and the backward pass below writes its own
        # order into it for the next frame
        ctx.order_slot = tile_order if isinstance(tile_order, list) else None
        if ctx.order_slot is not None:
            tile_order = ctx.order_slot[0] if ctx.order_slot[1] else None
        C, N = opac.shape
        dev = opac.device
        render = torch.empty(C, height, width, channels, dtype=torch.float32, device=dev)
        alpha = torch.empty(C, height, width, 1, dtype=torch.float32, device=dev)
        last_ids = torch.empty(C, height, width, dtype=torch.int32, device=dev)
        bg = _f32c(backgrounds, "backgrounds") if backgrounds is not None else None
        # per-tile work counts of this pass: the backward pass hands its tiles out costliest first (qed_composite_bwd)
        tile_cost = t_final = None
        if any(ctx.needs_input_grad[:5]):
            tile_cost = torch.empty(C * tile_w * tile_h, 4, dtype=torch.int32, device=dev)
            # the final transmittances themselves for the backward pass (1 - alpha keeps only ~3e-8 / T of them)
            if KEEP_T_FINAL:
                t_final = torch.empty(C, height, width, dtype=torch.float32, device=dev)
        # get_outputs' post-processing (model.py:295-297, 304-306) inside the same launch: rgb, depth as extra outputs
        post, post_rgb, post_depth, pbg = None, None, None, None
        if post_background is not None:
            pbg = _f32c(post_background, "background").reshape(3)
            post_rgb = torch.empty(C, height, width, 3, dtype=torch.float32, device=dev)
            post = L.Post()
            post.background, post.rgb = pbg.data_ptr(), post_rgb.data_ptr()
            if channels == 4:
                post_depth = torch.empty(C, height, width, 1, dtype=torch.float32, device=dev)
                dmax = torch.empty(C * tile_w * tile_h * 4, dtype=torch.float32, device=dev)
                post.depth, post.tile_dmax = post_depth.data_ptr(), dmax.data_ptr()
        L.check(lib.qed_composite_fwd(C, N, L.ptr(splats), L.ptr(flatten_ids), L.ptr(offsets), width, height, tile_w,
                                      tile_h, channels, L.ptr(bg), L.ptr(render), L.ptr(alpha), L.ptr(t_final),
                                      L.ptr(last_ids), L.ptr(tile_cost),
                                      L.ptr(tile_order) if (tile_order is not None and tile_order.numel() == C * tile_w * tile_h + 1)
                                      else None, C_byref(post), L.composite_launch_flags(), _stream()),
                "qed_composite_fwd")
        ctx.tile_cost = tile_cost
        ctx.save_for_backward(splats, flatten_ids, offsets, alpha, last_ids, bg, render if post is not None else None, pbg,
                              t_final)
        ctx.meta = (C, N, width, height, tile_w, tile_h, channels, absgrad)
        ctx.means2d_ref = means2d
        ctx.mark_non_differentiable(last_ids)
        if post is None:
            return render, alpha, last_ids
        if post_depth is None:
            return render, alpha, last_ids, post_rgb
        return render, alpha, last_ids, post_rgb, post_depth

    @staticmethod
    def backward(ctx, v_render, v_alpha, _v_last, v_rgb=None, v_depth=None):
        lib = L.load()
        splats, flatten_ids, offsets, alpha, last_ids, bg, render, pbg, t_final = ctx.saved_tensors
        C, N, width, height, tile_w, tile_h, channels, absgrad = ctx.meta
        dev = splats.device
        post = None
        if v_rgb is not None or v_depth is not None:
            v_rgb = _f32c(v_rgb, "v_rgb") if v_rgb is not None else None
            v_depth = _f32c(v_depth, "v_depth") if v_depth is not None else None
            if v_render is None and v_alpha is None:
                # the usual case on the reference-shaped route: only rgb / depth were used downstream -- their gradients go
                # straight into the compositing backward, which derives v_render / v_alpha per pixel in its tile prologue
                post = L.PostGrad()
                post.background, post.render = pbg.data_ptr(), render.data_ptr()
                post.v_rgb, post.v_depth = L.ptr(v_rgb), L.ptr(v_depth)
            else:
                # render / alpha were used as well: convert with the stand-alone pass and add
                vr2, va2 = torch.empty_like(render), torch.empty_like(alpha)
                L.check(lib.qed_post_process_bwd(C * height * width, channels, L.ptr(render), L.ptr(alpha), L.ptr(pbg),
                                                 L.ptr(v_rgb), L.ptr(v_depth), L.ptr(vr2), L.ptr(va2), _stream()),
                        "qed_post_process_bwd")
                v_render = vr2 if v_render is None else v_render + vr2
                v_alpha = va2 if v_alpha is None else v_alpha + va2
        if post is None:
            v_render = _f32c(v_render, "v_render") if v_render is not None else torch.zeros(
                C, height, width, channels, dtype=torch.float32, device=dev)
            v_alpha = _f32c(v_alpha, "v_alpha") if v_alpha is not None else torch.zeros(
                C, height, width, 1, dtype=torch.float32, device=dev)
        R = L.VSPLAT_FLOATS
        # the fused loss launch may already have zeroed an accumulator for this backward pass (model.fused_loss hands it
        # over through the holder; taken once -- a second backward through a retained graph makes its own)
        holder = ctx.vsplat_holder
        handed = holder.pop() if holder else None
        # (the fused loss launch hands over the zeroed accumulator and, when its SSIM pass carried the ordering job, the
        # launch order of THIS forward pass's tile costs: a dict; older callers hand over the accumulator alone)
        vsplat, order_ws = (handed.get("vsplat"), handed.get("order_ws")) if isinstance(handed, dict) else (handed, None)
        if vsplat is None or vsplat.shape != (C * N, R) or vsplat.device != dev or vsplat.dtype != torch.float32:
            vsplat = torch.zeros(C * N, R, dtype=torch.float32, device=dev)
        tile_cost = ctx.tile_cost
        flags = L.composite_launch_flags()
        if order_ws is not None and tile_cost is not None and order_ws.numel() == C * tile_w * tile_h + 1 and (flags & 3) == 0:
            flags |= L.CL_ORDER_READY
        elif tile_cost is not None:
            slot = ctx.order_slot
            if slot is not None and slot[0].numel() == C * tile_w * tile_h + 1 and (flags & 3) == 0:
                order_ws = slot[0]              # the camera's persistent buffer: the next frame's forward pass reads it
                slot[1] = True
            else:
                order_ws = torch.empty(C * tile_w * tile_h + 1, dtype=torch.int32, device=dev)
        else:
            order_ws = None
        L.check(lib.qed_composite_bwd(C, N, L.ptr(splats), L.ptr(flatten_ids), L.ptr(offsets), width, height, tile_w,
                                      tile_h, channels, L.ptr(bg), L.ptr(alpha), L.ptr(t_final), L.ptr(last_ids),
                                      L.ptr(v_render) if post is None else None, L.ptr(v_alpha) if post is None else None,
                                      L.ptr(vsplat), L.ptr(tile_cost), L.ptr(order_ws), C_byref(post), flags, _stream()),
                "qed_composite_bwd")
        ctx.order_ws = order_ws                  # (tests read the launch order: render.grad_fn.order_ws)
        v3 = vsplat.view(C, N, R)
        v_means2d = v3[..., 0:2]
        v_conics = v3[..., 4:7]
        v_opac = v3[..., 7]
        v_rgb_g = v3[..., 8:11]
        v_depths = v3[..., 11] if channels == 4 else None
        _VSPLAT_REGISTRY[vsplat.untyped_storage().data_ptr()] = vsplat
        leaf = ctx.grad_leaf
        if leaf is not None:
            # rasterization(_means2d_leaf=True): info["means2d"] is a leaf that takes the gradient as a VIEW of the
            # accumulator rows (what retain_grad() on the non-leaf clones: 4 MB through a strided copy kernel, ~9 us)
            leaf.grad = v_means2d
        if absgrad:
            # gsplat convention (absgrad=True at model.py:284): the densifier reads means2d.absgrad
            (leaf if leaf is not None else ctx.means2d_ref).absgrad = v3[..., 2:4]
        return (v_means2d, v_conics, v_rgb_g, v_opac, v_depths) + (None,) * 14


def C_byref(struct):
    """ctypes pointer to a structure argument (None -> NULL)."""
    import ctypes
    return None if struct is None else ctypes.cast(ctypes.pointer(struct), ctypes.c_void_p)


# ==================================================================================================
# public operator
# ==================================================================================================
def rasterization(
    means: Tensor, quats: Tensor, scales: Tensor, opacities: Tensor, colors: Tensor, viewmats: Tensor, Ks: Tensor,
    width: int, height: int, tile_size: int = 16, packed: bool = False, near_plane: float = 0.01,
    far_plane: float = 1e10, render_mode: str = "RGB", sh_degree: Optional[int] = None, sparse_grad: bool = False,
    absgrad: bool = False, rasterize_mode: str = "classic", radius_clip: float = 0.0, eps2d: float = 0.3,
    backgrounds: Optional[Tensor] = None, _flags: int = 0, _sh_rest: Optional[Tensor] = None,
    _sync: bool = True, _vsplat_holder: Optional[list] = None, _c2w: Optional[Tuple[Tensor, Tensor]] = None,
    _post_background: Optional[Tensor] = None, _means2d_leaf: bool = False, _capture_slot=None,
    _manual: Optional[list] = None, _lazy_sh=None, _tile_order: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor, Dict]:
    """Same call surface as the reference's call (model.py:267-288).

    ``_flags`` / ``_sh_rest`` are the fused entry used by ``qed_splatter_amd.model``: with them
    ``scales`` / ``opacities`` may be raw log-scales / logits (QED_F_LOG_SCALES / QED_F_LOGIT_OPAC)
    and ``colors`` / ``_sh_rest`` may be features_dc / features_rest without the torch.cat of
    model.py:241.  ``_post_background`` [3]: the statements that follow the call in get_outputs (model.py:295-297,
    304-306) run inside the compositing kernels; ``info["post_rgb"]`` [C,H,W,3] and ``info["post_depth"]`` [C,H,W,1]
    (RGB+D) are their results, differentiable like ``render`` / ``alpha``.  ``_means2d_leaf``: ``info["means2d"]`` is a
    leaf that receives ``.grad`` / ``.absgrad`` as views (see below) instead of gsplat's non-leaf.
    """
    if packed or sparse_grad:
        raise NotImplementedError("packed=True / sparse_grad=True are not used by the reference (model.py:278,283)")
    if tile_size != L.TILE:
        raise NotImplementedError("tile_size must be 16 (BLOCK_WIDTH, model.py:243)")
    if render_mode not in ("RGB", "RGB+D"):
        raise NotImplementedError(f"render_mode {render_mode!r}: the reference uses 'RGB' / 'RGB+D' (model.py:256-259)")
    if rasterize_mode not in ("classic", "antialiased"):
        raise ValueError(f"Unknown rasterize_mode: {rasterize_mode}")
    if not means.is_cuda:
        raise L.QedSplatError("rasterization() needs GPU tensors: there is no CPU path in the product")
    N = means.shape[0]
    C = viewmats.shape[0]
    assert means.shape == (N, 3) and quats.shape == (N, 4) and scales.shape == (N, 3)
    assert opacities.numel() == N and viewmats.shape == (C, 4, 4) and Ks.shape == (C, 3, 3)
    tile_w = math.ceil(width / tile_size)
    tile_h = math.ceil(height / tile_size)

    flags = int(_flags)
    if rasterize_mode == "antialiased":
        flags |= L.F_ANTIALIASED
    channels = 3
    if render_mode == "RGB+D":
        flags |= L.F_DEPTH_CHANNEL
        channels = 4
    if sh_degree is None:
        deg = -1
        if colors.dim() != 2 or colors.shape != (N, 3):
            raise NotImplementedError("sh_degree=None needs colors [N,3] (model.py:263-265)")
        sh0, shN = colors, None
    else:
        deg = int(sh_degree)
        if deg > 3:
            raise NotImplementedError("SH degree > 3 (the reference config uses sh_degree=3)")
        if _sh_rest is not None:
            sh0, shN = colors.reshape(N, 3), _sh_rest
            assert shN.shape[1] >= (deg + 1) ** 2 - 1
        else:
            assert colors.dim() == 3 and colors.shape[0] == N and colors.shape[2] == 3
            assert colors.shape[1] >= (deg + 1) ** 2, "colors must hold (sh_degree+1)^2 coefficients"
            sh0, shN = colors, None

    # (_manual: a list that receives the two nodes' contexts -- the call then runs without autograd, manual_backward()
    # is its backward pass: what a captured segment replays)
    means2d, depths, conics, opac, rgb, radii, splats, tiles_per_gauss, block_sums, tile_masks = _run_node(
        _ProjectSH, _manual, None,
        means, quats, scales, opacities, sh0, shN, viewmats, Ks, int(width), int(height), tile_w, tile_h, deg, flags,
        float(eps2d), float(near_plane), float(far_plane), float(radius_clip),
        *(_c2w if _c2w is not None else (None, None)), _lazy_sh)

    # the packed tile rectangles of the records save the emit pass a recomputation; with F_TIGHT_TILES they
    # are the only place the (smaller) rectangles exist
    use_packed = tile_w <= 1023 and tile_h <= 2047
    if (flags & L.F_TIGHT_TILES) and not use_packed:
        raise NotImplementedError("F_TIGHT_TILES needs tile grids of at most 1023 x 2047 tiles")
    isect_ids, flatten_ids, offsets, M = _bin_and_sort(N, C, means2d, radii, depths, tiles_per_gauss, block_sums,
                                                       tile_w, tile_h, sync=_sync, splats=splats if use_packed else None,
                                                       size=(int(width), int(height)),
                                                       tile_masks=tile_masks if tile_masks.numel() else None,
                                                       capture_slot=_capture_slot)
    # _means2d_leaf: hand out info["means2d"] as a LEAF holding the same values, which receives .grad / .absgrad from
    # the compositing backward without a copy.  For callers that only retain and read the gradient (the reference:
    # model.py:289-290 and the densification strategy); a loss computed FROM info["means2d"] would not reach the
    # Gaussians through the leaf, which is why it is not the default.
    means2d_out, grad_leaf = means2d, None
    wants_grad = means2d.requires_grad or (_manual is not None and any(_manual[0].needs_input_grad[:6]))
    if _means2d_leaf and wants_grad:
        means2d_out = means2d.detach().requires_grad_(True)
        grad_leaf = [means2d_out]
    outs = _run_node(_Composite, _manual, (wants_grad,) * 5 + (False,) * 14,
                     means2d, conics, rgb, opac, depths if channels == 4 else None, splats, flatten_ids, offsets,
                     backgrounds, int(width), int(height), tile_w, tile_h, channels, bool(absgrad), _vsplat_holder,
                     _post_background, grad_leaf, _tile_order)
    render, alpha, last_ids = outs[:3]
    info = {
        "camera_ids": None, "gaussian_ids": None,
        "radii": radii, "means2d": means2d_out, "depths": depths, "conics": conics, "opacities": opac,
        "tile_width": tile_w, "tile_height": tile_h, "tiles_per_gauss": tiles_per_gauss,
        "isect_ids": isect_ids, "flatten_ids": flatten_ids,
        "isect_offsets": offsets[: C * tile_w * tile_h].view(C, tile_h, tile_w),
        "width": width, "height": height, "tile_size": tile_size, "n_cameras": C,
        "last_ids": last_ids, "colors": rgb, "n_isects": M,
    }
    if _post_background is not None:
        info["post_rgb"] = outs[3]
        info["post_depth"] = outs[4] if channels == 4 else None
    return render, alpha, info
