"""GPU tests of the drop-in route under the REFERENCE'S DEFAULT CONFIGURATION: the coarse-to-fine resolution schedule
(parent defaults num_downscales=2 / resolution_schedule=3000 behind /root/reference/qed_splatter/model.py:244-250), one
optimiser per parameter group (config.py:44-68) and densification (config.py:40-41) -- with the intersection count
left on the device (``async_intersection_count``).  The intersection buffer is calibrated per (width, height, N, C);
a frame that overflows it all the same must neither be trained on nor raise a step late."""
from __future__ import annotations

import math
import warnings

import pytest
import torch

from tests.util import PARAM_NAMES, scene

pytestmark = pytest.mark.gpu


def _big_splat_scene(n, w, h, seed, grow=math.log(10.0)):
    """The synthetic scene with ten times larger Gaussians: tens of tiles per Gaussian at full resolution (real
    captures sit well above the 7-9 tiles per Gaussian of SURVEY 8d's scene) and a learnable, constant ground truth."""
    sc = scene(n, w, h, seed=seed)
    sc["scales"] = sc["scales"] + grow
    sc["opacities"] = sc["opacities"] - 3.0           # faint: the gradient reaches the Gaussians behind the front ones
    sc["gt_rgb"] = torch.full((h, w, 3), 0.5)
    sc["gt_depth"] = torch.full((h, w, 1), 6.0)
    return sc


def _setup(sc, dev, **cfg_kw):
    from qed_splatter_amd.model import PinholeCameras, QedAdam, QEDSplatterModel, QEDSplatterModelConfig
    cfg_kw.setdefault("sh_degree_interval", 1)
    cfg_kw.setdefault("background_color", "black")
    cfg = QEDSplatterModelConfig(**cfg_kw)
    m = QEDSplatterModel(cfg, **{k: sc[k].to(dev) for k in PARAM_NAMES})
    K = sc["Ks"][0]
    h, w = sc["gt_rgb"].shape[:2]
    cam = PinholeCameras(sc["camera_to_worlds"][:1].to(dev), K[0, 0], K[1, 1], K[0, 2], K[1, 2], w, h)
    batch = {"image": sc["gt_rgb"].to(dev), "depth_image": sc["gt_depth"].to(dev)}
    lrs = {"means": 1.6e-4, "scales": 0.005, "quats": 0.001, "opacities": 0.05, "features_dc": 0.0025,
           "features_rest": 0.0025 / 20}                                               # config.py:44-68
    opts = {n: QedAdam([m.gauss_params[n]], lr=lrs[n], eps=1e-15) for n in m.group_names}
    return m, cam, batch, opts


def _api_step(m, cam, batch, opts, step):
    """The trainer's call sequence for one iteration (Nerfstudio's Trainer.train_iteration around model.py:199-321 and
    :73-118): outputs -> metrics -> loss dict -> sum -> backward -> one optimiser per group."""
    m.step = step
    for o in opts.values():
        o.zero_grad()
    out = m.get_outputs(cam)
    m.get_metrics_dict(out, batch)
    ld = m.get_loss_dict(out, batch)
    loss = sum(ld.values())
    loss.backward()
    for o in opts.values():
        o.step()
    return out, loss.detach()


def test_async_route_survives_resolution_schedule_and_refinement(cuda):
    """ONE model in training mode through both resolution steps and a refinement that multiplies the intersections,
    starting from a workspace that has seen nothing: every frame is drawn, nothing raises, the loss falls."""
    from qed_splatter_amd import rasterization as R
    from qed_splatter_amd.densify import DensifyConfig, Densifier
    from qed_splatter_amd.model import QedAdamSet
    R._WORKSPACES.clear()
    n, w, h = 6_000, 640, 480
    sc = _big_splat_scene(n, w, h, seed=31)
    m, cam, batch, opts = _setup(sc, cuda, num_downscales=2, resolution_schedule=4)
    assert m.config.async_intersection_count
    m.train()
    # every visible Gaussian is duplicated at the refinement (nothing is "large", every gradient is above the threshold)
    dcfg = DensifyConfig(warmup_length=5, refine_every=5, densify_grad_thresh=1e-12, densify_size_thresh=1e3,
                         cull_alpha_thresh=1e-6, cull_scale_thresh=1e3, cull_screen_size=1e3, split_screen_size=1e3)
    dens = Densifier(m, QedAdamSet(m, opts), dcfg, num_train_data=1, seed=0)
    acc, losses, sizes, lists, counts = [], [], [], [], []
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)          # an overflow warning would be a failure here
        for step in range(16):
            out, loss = _api_step(m, cam, batch, opts, step)
            acc.append(out["accumulation"].sum())
            losses.append(loss)
            sizes.append(tuple(out["rgb"].shape[:2]))
            lists.append(m.info["tiles_per_gauss"].sum())
            counts.append(m.num_points)
            dens.after_train(step)
            if step == 10:
                info = dens.refinement_after(step)
                assert info["did_densify"] and info["n_after"] > 1.25 * info["n_before"], info
    torch.cuda.synchronize()
    ws = R._workspace(cuda)
    ws.poll_pending()
    assert ws.overflows == 0
    acc = [float(a) for a in acc]
    lists = [int(x) for x in lists]
    assert sizes[0] == (h // 4, w // 4) and sizes[4] == (h // 2, w // 2) and sizes[8] == (h, w), sizes
    assert min(acc) > 0.0, acc                                                     # no frame rendered empty
    # the schedule and the refinement did what the old capacity rule could not survive
    assert lists[8] > 8 * n, (lists[8], n)                                         # M beyond 8 x N_initial
    assert lists[11] > 1.25 * lists[10] and counts[11] > 1.25 * counts[10], (lists[10:12], counts[10:12])
    assert lists[8] > 1.8 * lists[7] and lists[4] > 1.8 * lists[3], lists
    ls = [float(x) for x in losses]
    assert all(math.isfinite(x) for x in ls)
    assert sum(ls[-3:]) < sum(ls[:3]), ls                                          # the loss falls


def test_async_overflow_skips_the_update_warns_and_recovers(cuda):
    """Force what the headroom makes rare: an asynchronous frame that needs more room than the buffer has.  That frame
    renders empty; the optimiser launches behind it are no-ops on the device (bit-identical parameters and moments);
    the next call warns, regrows, reads M back and training goes on -- nothing raises."""
    from qed_splatter_amd import rasterization as R
    R._WORKSPACES.clear()
    n, w, h = 8_000, 320, 240
    sc = _big_splat_scene(n, w, h, seed=32, grow=math.log(4.0))
    m, cam, batch, opts = _setup(sc, cuda, num_downscales=0)
    m.train()
    for step in range(3):                                       # calibrates (first call) and goes asynchronous
        out, _ = _api_step(m, cam, batch, opts, step)
    torch.cuda.synchronize()
    ws = R._workspace(cuda)
    ws.poll_pending()
    M = int(m.info["tiles_per_gauss"].sum())
    assert ws.capacity >= 2 * M > 0 and ws.pending is None
    ws.capacity = max(M // 3, 1024)                             # (what a sudden change of the scene would amount to)
    before = m.flat_params.clone()
    st = opts["means"]._shared
    m_before, v_before = st.exp_avg.clone(), st.exp_avg_sq.clone()
    out, _ = _api_step(m, cam, batch, opts, 3)                  # overflows: empty frame, skipped update
    torch.cuda.synchronize()
    assert float(out["accumulation"].sum()) == 0.0
    assert torch.equal(m.flat_params, before)
    assert torch.equal(st.exp_avg, m_before) and torch.equal(st.exp_avg_sq, v_before)
    with pytest.warns(RuntimeWarning, match="rendered empty"):
        out, _ = _api_step(m, cam, batch, opts, 4)              # regrown, synchronous, trained on
    torch.cuda.synchronize()
    assert ws.overflows == 1 and m.intersection_overflows == 1 and ws.capacity >= 2 * M and int(ws.status[0]) == 0
    assert float(out["accumulation"].sum()) > 0.0
    assert not torch.equal(m.flat_params, before)
    # five step() calls, four updates: the host step counters (bias corrections, torch-layout checkpoints) were told
    for name in m.group_names:
        o = opts[name]
        assert o._shared.t[m.gauss_params[name].storage_offset()] == 4, name
        assert float(o.state[m.gauss_params[name]]["step"]) == 4.0, name
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        for step in range(5, 8):                                # and asynchronous again
            out, _ = _api_step(m, cam, batch, opts, step)
        assert ws.pending is not None
    assert float(out["accumulation"].sum()) > 0.0


def test_overflow_is_visible_to_a_trainer_that_steps_torch_adam(cuda):
    """The reference's own optimisers (torch.optim.Adam per group, config.py:44-68) do not take the device-side skip word:
    an empty (overflowed) frame's zero gradients would still move the parameters by the momentum.  The trainer can see the
    overflow in time -- ``model.frame_overflowed()`` between backward and the steps -- and drop the step; the frame after
    carries ``info["intersection_overflow_previous_frame"]`` (the dict the reference keeps as self.info, model.py:267)."""
    from qed_splatter_amd import rasterization as R
    R._WORKSPACES.clear()
    n, w, h = 8_000, 320, 240
    sc = _big_splat_scene(n, w, h, seed=33, grow=math.log(4.0))
    m, cam, batch, _ = _setup(sc, cuda, num_downscales=0, lazy_sh_grad=False)
    lrs = {"means": 1.6e-4, "scales": 0.005, "quats": 0.001, "opacities": 0.05, "features_dc": 0.0025,
           "features_rest": 0.0025 / 20}
    opts = {k: torch.optim.Adam([m.gauss_params[k]], lr=lrs[k], eps=1e-15) for k in m.group_names}
    m.train()

    def iteration(step):
        m.step = step
        for o in opts.values():
            o.zero_grad()
        out = m.get_outputs(cam)
        flag_prev = m.info["intersection_overflow_previous_frame"]
        sum(m.get_loss_dict(out, batch).values()).backward()
        dropped = m.frame_overflowed()
        if not dropped:
            for o in opts.values():
                o.step()
        return out, flag_prev, dropped

    for step in range(3):
        out, flag_prev, dropped = iteration(step)
        assert not flag_prev and not dropped
    torch.cuda.synchronize()
    ws = R._workspace(cuda)
    M = int(m.info["tiles_per_gauss"].sum())
    ws.capacity = max(M // 3, 1024)
    before = m.flat_params.clone()
    mom = {k: opts[k].state[m.gauss_params[k]]["exp_avg"].clone() for k in m.group_names}
    steps_before = {k: float(opts[k].state[m.gauss_params[k]]["step"]) for k in m.group_names}
    with pytest.warns(RuntimeWarning, match="rendered empty"):
        out, flag_prev, dropped = iteration(3)                  # overflows; the trainer sees it and drops the step
    torch.cuda.synchronize()
    assert dropped and not flag_prev and float(out["accumulation"].sum()) == 0.0
    assert torch.equal(m.flat_params, before)
    for k in m.group_names:
        st = opts[k].state[m.gauss_params[k]]
        assert torch.equal(st["exp_avg"], mom[k]) and float(st["step"]) == steps_before[k], k
    out, flag_prev, dropped = iteration(4)                      # regrown, synchronous, trained on; the dict says what happened
    torch.cuda.synchronize()
    assert flag_prev and not dropped and float(out["accumulation"].sum()) > 0.0
    assert not torch.equal(m.flat_params, before)
    out, flag_prev, dropped = iteration(5)
    assert not flag_prev and not dropped


def test_graphed_step_overflow_is_skipped_and_recaptured(cuda):
    """The same guarantee for a replayed hipGraph, whose buffers cannot grow: replays that overflow train nothing,
    check() re-captures with room and returns False, the replays after it train."""
    from qed_splatter_amd import rasterization as R
    from qed_splatter_amd.graph import GraphedTrainStep
    from qed_splatter_amd.model import FlatAdam
    R._WORKSPACES.clear()
    n, w, h = 8_000, 320, 240
    sc = _big_splat_scene(n, w, h, seed=33, grow=0.0)
    m, cam, batch, _ = _setup(sc, cuda, num_downscales=0)
    m.train()
    m.step = 10
    opt = FlatAdam(m)

    def step():
        for p in m.gauss_params.values():
            p.grad = None
        losses = m.fused_loss(cam, batch, sync=False, compact_sh_grad=True, optimizer=opt)
        m.backward_fused(losses)
        opt.step(device_state=True, fused_sh=True)
        return losses

    g = GraphedTrainStep(step, cuda, warmup=2, check_every=0)
    g.replay()
    assert g.check() is True
    ws = R._workspace(cuda)
    cap = ws.capacity
    with torch.no_grad():
        m.scales.add_(math.log(30.0))                           # the scene changes under the captured graph
    torch.cuda.synchronize()
    before, t_before = m.flat_params.clone(), opt.dev_state.clone()
    g.replay()
    g.replay()
    torch.cuda.synchronize()
    assert int(ws.status[0]) > cap                              # the list no longer fits the captured buffers
    assert torch.equal(m.flat_params, before) and torch.equal(opt.dev_state, t_before)
    with pytest.warns(RuntimeWarning, match="Re-captured"):
        assert g.check() is False
    assert ws.capacity > cap and int(ws.status[0]) == 0
    out = g.replay()
    torch.cuda.synchronize()
    assert g.check() is True
    assert math.isfinite(float(out["loss"])) and not torch.equal(m.flat_params, before)
