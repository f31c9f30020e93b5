"""hipGraph capture of one whole training step (forward + fused loss + backward + Adam).

The step is ~45 short kernels; eager PyTorch dispatch costs more host time than the GPU needs to run
them.  Every kernel behind the C ABI reads data-dependent sizes (the intersection count M) from
device memory and sizes its grid by a calibrated capacity, and the Adam step keeps its counter in
device memory (qed_adam_step_dev), so the step can be captured once and replayed.  Inputs are
static tensors: copy a new camera / ground truth into them before ``replay()``.
"""
from __future__ import annotations

from typing import Callable, Dict

import torch

from . import _lib as L
from .rasterization import _workspace


class GraphedTrainStep:
    """Capture ``step_fn`` (any callable that runs one full step on static tensors and returns a dict
    of scalar loss tensors) after a few eager warm-up runs on a side stream; ``replay()`` re-runs it.

    ``check_every``: every that many replays the intersection-overflow word is read back (one small
    D2H copy).  A frame that overflowed the captured buffer rendered empty and the optimiser launches of
    that and the following replays were no-ops on the device (``skip_flag`` of ``qed_adam_step*``);
    ``check()`` then makes room, captures again, warns and returns False -- nothing is trained on an
    empty frame and nothing raises a step late.
    """

    def __init__(self, step_fn: Callable[[], Dict[str, torch.Tensor]], device, warmup: int = 3, check_every: int = 50):
        self.device = torch.device(device)
        self.step_fn = step_fn
        self.check_every = check_every
        self.warmup = warmup
        self.n_replays = 0
        self._capture()

    def recapture(self) -> None:
        """Capture again after anything the captured launches were specialised on has changed: the number
        of Gaussians (densification swaps the flat parameter / moment buffers), the image size, the SH degree."""
        self.graph = None
        self._capture()

    def _capture(self) -> None:
        ws = _workspace(self.device)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(max(self.warmup, 1)):
                self.step_fn()                  # also calibrates the intersection capacity (first call syncs)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        ws.poll_pending()
        # (the captured buffers can never grow: the workspace keeps _Workspace.HEADROOM x the longest list seen)
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: HIP calls made by other threads (e.g. the RCCL watchdog polling its events in a
        # data-parallel job) must not invalidate this thread's capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.outputs = self.step_fn()
        self.ws = ws

    def replay(self) -> Dict[str, torch.Tensor]:
        self.graph.replay()
        self.n_replays += 1
        if self.check_every and self.n_replays % self.check_every == 0:
            self.check()
        return self.outputs

    def check(self) -> bool:
        """True when every replay since the last check had room for its list.  Otherwise: grow, re-capture, warn."""
        import warnings
        status = self.ws.status.tolist()        # synchronises
        if status[1]:
            self.ws.status.zero_()
            raise L.QedSplatError("radix-sort look-back watchdog fired")
        if status[0]:
            need, old = int(status[0]), self.ws.capacity
            self.ws.status.zero_()
            self.ws.overflows += 1
            self.ws.capacity = max(self.ws.capacity, int(need * self.ws.HEADROOM) + 4096)
            self.ws.force_sync = True           # the warm-up call of the re-capture reads M back
            warnings.warn(f"qed_splatter_amd: a graphed step needed {need} tile intersections, more than the captured "
                          f"capacity {old}; the frames since rendered empty and their optimiser steps were skipped on "
                          f"the device.  Re-captured with capacity {self.ws.capacity}.", RuntimeWarning, stacklevel=2)
            self.recapture()
            return False
        return True
