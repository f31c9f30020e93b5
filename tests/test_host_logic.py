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
