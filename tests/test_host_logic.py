"""Host-side logic that needs no GPU: when get_outputs' captured segments are worth their capture (segments.SegmentCache)."""
from __future__ import annotations

import os
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


def test_lazy_sh_gradient_parameters_complete_the_gradients_for_every_python_reader(monkeypatch):
    """model._LazySHGradParameter (config.lazy_sh_grad): while the model holds a compact SH gradient, READING ``.grad`` of
    features_dc / features_rest first has the full gradients written (here: a stand-in that counts), ``.grad = None`` on both
    drops the compact form, assigning a tensor completes the other field first, and the four geometry groups are plain
    Parameters.  The device side of it is tests/test_api_path.py::test_lazy_sh_gradients_*."""
    import copy
    import torch
    from qed_splatter_amd.model import QEDSplatterModel, _LazySHGradParameter, _raw_grad
    from tests.util import PARAM_NAMES, scene
    sc = scene(40, 32, 32, seed=3)
    m = QEDSplatterModel(None, **{k: sc[k] for k in PARAM_NAMES})
    dc, rest = m.gauss_params["features_dc"], m.gauss_params["features_rest"]
    assert type(dc) is _LazySHGradParameter and type(rest) is _LazySHGradParameter
    assert all(type(m.gauss_params[k]) is torch.nn.Parameter for k in ("means", "scales", "quats", "opacities"))
    assert dc.data_ptr() == m.flat_params.data_ptr() + 4 * m.group_begin[4]           # still views of the flat buffer
    calls = []

    def fake_materialise():
        calls.append(1)
        m.__dict__["_lazy_sh"] = None
    monkeypatch.setattr(m, "_materialise_sh_grads", fake_materialise)

    def pend():
        _raw_grad_set(dc, torch.zeros_like(dc))
        _raw_grad_set(rest, torch.zeros_like(rest))
        m.__dict__["_lazy_sh"] = {"stand-in": True}

    from qed_splatter_amd.model import _RAW_GRAD
    _raw_grad_set = _RAW_GRAD.__set__
    # no compact form: plain behaviour
    assert dc.grad is None and not calls
    # a read completes the gradients, once
    pend()
    assert rest.grad is not None and len(calls) == 1
    assert dc.grad is not None and len(calls) == 1
    # the raw accessor (what QedAdam uses) does not
    pend()
    assert _raw_grad(dc) is not None and len(calls) == 1
    # .grad = None on both drops the compact form without completing it
    dc.grad = None
    assert m.__dict__["_lazy_sh"] is not None and len(calls) == 1
    rest.grad = None
    assert m.__dict__["_lazy_sh"] is None and len(calls) == 1
    # assigning a tensor to one field completes the other first
    pend()
    dc.grad = torch.ones_like(dc)
    assert len(calls) == 2 and float(_raw_grad(dc).sum()) == dc.numel()
    # nn.Module.zero_grad and torch.optim see ordinary Parameters
    pend()
    m.zero_grad()
    assert len(calls) == 3 and _raw_grad(dc) is None and _raw_grad(rest) is None
    # a deep copy is a Parameter of the same class without an owner: plain behaviour
    c = copy.deepcopy(dc)
    assert isinstance(c, torch.nn.Parameter) and c.grad is None
    # after densification (rebind_flat) the new Parameters are lazy ones again and nothing is pending
    m.__dict__["_lazy_sh"] = {"stand-in": True}
    m.rebind_flat(m.flat_params.detach().clone(), m.num_points)
    assert m.__dict__["_lazy_sh"] is None and type(m.gauss_params["features_rest"]) is _LazySHGradParameter


def test_committed_counters_are_quoted_only_for_the_sources_they_were_taken_on(tmp_path):
    """bench.py's roofline.traffic / roofline_issue.frac come from committed --pmc summaries: a summary is quoted only
    while the kernel sources it records (sha256) are the tree's; an edited kernel, a summary without the record or a
    missing file give None (the bench line then carries traffic: null)."""
    import hashlib
    import json
    import bench

    src = tmp_path / "qed_splatter_amd" / "csrc"
    src.mkdir(parents=True)
    (tmp_path / "profiles").mkdir()
    (src / "composite.hip").write_text("// kernel v1\n")
    h = hashlib.sha256((src / "composite.hip").read_bytes()).hexdigest()
    body = {"kernels_version": "t", "source_sha256": {"qed_splatter_amd/csrc/composite.hip": h},
            "kernels": {"qed::composite_bwd_kernel<4>": {"fetch_size_kb": 1.0, "write_size_kb": 2.0}}}
    (tmp_path / "profiles" / "good.json").write_text(json.dumps(body))
    (tmp_path / "profiles" / "nohash.json").write_text(json.dumps({k: v for k, v in body.items() if k != "source_sha256"}))
    (tmp_path / "profiles" / "empty.json").write_text(json.dumps({**body, "source_sha256": {}}))
    assert bench.committed_counters("good.json", root=str(tmp_path))["kernels"]
    assert bench.committed_counters("nohash.json", root=str(tmp_path)) is None
    assert bench.committed_counters("empty.json", root=str(tmp_path)) is None
    assert bench.committed_counters("absent.json", root=str(tmp_path)) is None
    (src / "composite.hip").write_text("// kernel v2\n")                 # the kernel changes: the summary goes stale
    assert bench.committed_counters("good.json", root=str(tmp_path)) is None
    # the summaries bench.py names are either absent (not taken yet for this tree) or match it -- never silently stale
    for name in (bench.PMC_TRAFFIC, bench.PMC_VALU):
        path = os.path.join(bench.ROOT, "profiles", name)
        if os.path.exists(path) and bench.committed_counters(name) is None:
            import warnings
            warnings.warn(f"{name} does not describe this tree's kernels: bench.py will report traffic: null")


def test_lazy_sh_gradients_are_off_whenever_a_c_level_reader_could_see_them(monkeypatch):
    """Only Python reads of ``.grad`` complete a compact SH gradient.  DistributedDataParallel's reducer and tensor /
    post-accumulate-grad hooks take the raw field, so with more than one rank in the default process group, or with a hook
    on either SH Parameter, a training step writes its gradients out."""
    import types as _types
    import torch
    from qed_splatter_amd import model as M
    from tests.util import PARAM_NAMES, scene
    sc = scene(40, 32, 32, seed=3)
    m = M.QEDSplatterModel(None, **{k: sc[k] for k in PARAM_NAMES})
    m.train()
    # stand-in for "all six groups are stepped by QedAdam instances" (the registry itself needs GPU tensors)
    class _Six:                                                          # (WeakValueDictionary: needs a weak-referenceable value)
        members = [0] * 6
    six = _Six()
    monkeypatch.setitem(M._FLAT_STATES, m._flat.untyped_storage().data_ptr(), six)
    assert m._lazy_sh_wanted(3, None) is True
    assert m._lazy_sh_wanted(None, None) is False                       # sh_degree 0 in use: colours, no coefficients
    monkeypatch.setattr(M, "_dist_world_size", lambda: 2)
    assert m._lazy_sh_wanted(3, None) is False
    monkeypatch.setattr(M, "_dist_world_size", lambda: 1)
    assert m._lazy_sh_wanted(3, None) is True
    h = m.gauss_params["features_rest"].register_hook(lambda g: g)
    assert m._lazy_sh_wanted(3, None) is False
    h.remove()
    assert m._lazy_sh_wanted(3, None) is True
    h = m.gauss_params["features_dc"].register_post_accumulate_grad_hook(lambda p: None)
    assert m._lazy_sh_wanted(3, None) is False
    h.remove()
    assert m._lazy_sh_wanted(3, None) is True
