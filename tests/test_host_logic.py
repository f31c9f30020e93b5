"""Host-side logic that needs no GPU: when get_outputs' captured segments are worth their capture (segments.SegmentCache)."""
from __future__ import annotations

import types

import pytest


def test_segment_capture_policy_is_ski_rental_and_host_bound_only(monkeypatch):
    """config.graph_segments = True captures a shape only (i) after it has been called as often as a capture costs in
    per-call savings -- a shape that densification replaces every 100 steps is never captured -- and (ii) when the host, not
    the device, is the slower side (it waited for the device less than a tenth of the time); "always" captures on the
    fourth call; a shape change restarts the count."""
    import time

    from qed_splatter_amd.segments import SegmentCache
    clock = [0.0]
    monkeypatch.setattr(time, "perf_counter", lambda: clock[0])
    ws = types.SimpleNamespace(waited_s=0.0)

    def run(cache, key, mode, calls, step_s, wait_s):
        """Index of the first call that says 'capture now' (None: never within `calls`)."""
        for i in range(calls):
            clock[0] += step_s
            ws.waited_s += wait_s
            if cache.should_capture(key, mode, ws):
                return i + 1
        return None

    c = SegmentCache()
    assert run(c, "a", "always", 10, 1e-3, 5e-4) == SegmentCache.WARM_CALLS + 1
    need = int(c.capture_s / SegmentCache.SAVING_S)                       # ~266 calls for a 40 ms capture
    assert 100 < need < 1000
    # host-bound (waits 2 % of the time): captured right after the rental period
    c = SegmentCache()
    assert run(c, "a", True, 2 * need, 1e-3, 2e-5) == need + 1
    # device-bound (waits 30 %): never captured, and the reason is recorded
    c = SegmentCache()
    assert run(c, "a", True, 5 * need, 1e-3, 3e-4) is None
    assert "waited for the device 30 %" in c.decision
    # ... until the host becomes the slower side: captured one rental period later
    assert run(c, "a", True, 3 * need, 1e-3, 1e-5) is not None
    # densification: a new shape every 100 calls never lives long enough
    c = SegmentCache()
    for gen in range(12):
        assert run(c, ("n", gen), True, 100, 1e-3, 0.0) is None
    # a capture that took 5 ms shortens the rental period accordingly
    c = SegmentCache()
    c.capture_s = 5e-3
    assert run(c, "b", True, 1000, 1e-3, 0.0) == int(5e-3 / SegmentCache.SAVING_S) + 1
    # once capturing failed for a model it stays eager
    c = SegmentCache()
    c.disabled = "RuntimeError: x"
    assert run(c, "a", "always", 50, 1e-3, 0.0) is None


def test_step_context_hands_everything_out_once_and_only_to_its_own_outputs():
    """model.StepContext (CPU: plain tensors): the batch conversion is memoised by source identity + version + downscale
    factor; the SSIM forward, the accumulator hand-over and a segment's static gradient buffers are each handed out ONCE;
    `owns` is an identity test on outputs["rgb"]."""
    import torch

    from qed_splatter_amd.model import StepContext
    ctx = StepContext()
    rgb = torch.zeros(4, 5, 3)
    assert not ctx.owns({"rgb": rgb})                              # nothing bound yet
    holder: list = []
    ctx.bind(rgb, holder, 7)
    assert ctx.owns({"rgb": rgb}) and not ctx.owns({"rgb": rgb.clone()}) and not ctx.owns({})
    # conversions
    calls = []

    def convert(img):
        calls.append(1)
        return img.float() / 255.0
    img = torch.full((4, 5, 3), 255, dtype=torch.uint8)
    a = ctx.gt_image(img, 1, convert)
    b = ctx.gt_image(img, 1, convert)
    assert a is b and len(calls) == 1                              # same source, same version, same factor
    ctx.gt_image(img, 2, convert)
    assert len(calls) == 2                                         # another downscale factor
    img.add_(0)                                                    # in-place write: new version
    ctx.gt_image(img, 2, convert)
    assert len(calls) == 3
    ctx.gt_image(img.clone(), 2, convert)
    assert len(calls) == 4                                         # another tensor
    # one-shot hand-overs
    ctx.ssim = {"key": 1}
    assert ctx.take_ssim() == {"key": 1} and ctx.take_ssim() is None
    assert ctx.take_grad_buffers() is None                         # no captured segment behind these outputs
    acc = ctx.take_accumulator()
    assert acc[0] is holder and acc[1] == 7 and acc[2] is None and ctx.take_accumulator() is None
    # with a segment's static buffers
    ctx2 = StepContext()
    ctx2.bind(rgb, holder, 7)
    v_rgb, v_depth, vsplat = torch.zeros(4, 5, 3), torch.zeros(4, 5, 1), torch.zeros(7, 16)
    ctx2.static = (v_rgb, v_depth, vsplat)
    got = ctx2.take_grad_buffers()
    assert got[0] is v_rgb and got[1] is v_depth and ctx2.take_grad_buffers() is None
    assert ctx2.take_accumulator()[2] is vsplat                    # the accumulator survives the gradient buffers' hand-over
