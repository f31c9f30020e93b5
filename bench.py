#!/usr/bin/env python3
"""bench.py -- the hot-path benchmark (contract in the round prompt / BASELINE.json).

One "step" = one training iteration of the depth-supervised splatting hot path on one camera:
projection+SH -> tile binning -> radix sort -> compositing forward -> fused L1-RGB + depth-L1 loss
-> compositing backward -> projection+SH backward -> (N > 1: RCCL all-reduce of the flat gradient)
-> fused Adam step.  Workload at N = 1: BASELINE.json configs[1] = 500k Gaussians, SH degree 3,
1 camera @ 1920x1080, synthetic scene of SURVEY.md section 8(d), resident in HBM before timing.

N > 1 (launched by torch.distributed.run): every rank holds the full Gaussian set and renders its
own camera (yawed by 5 degrees * rank); gradients are summed with one all-reduce of the flat 59 N
float buffer; weak scaling (per-GPU work fixed).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# committed rocprofv3 --pmc summaries the roofline objects quote (re-taken whenever the compositing kernels change)
PMC_TRAFFIC = "r05_hbm_traffic_pmc.json"
PMC_VALU = "r05_valu_issue_pmc.json"

import torch  # noqa: E402


def committed_counters(name, root=ROOT):
    """A committed rocprofv3 --pmc summary under profiles/, or None when it does not describe THIS tree's kernels: the
    summary records sha256 of the kernel sources it was taken on (scripts/pmc_to_json.py: `source_sha256`), and a
    summary without that record, or whose sources have changed since, is not quoted."""
    import hashlib
    try:
        prof = json.load(open(os.path.join(root, "profiles", name)))
        hashes = prof["source_sha256"]
        if not hashes:
            return None
        for rel, want in hashes.items():
            if hashlib.sha256(open(os.path.join(root, rel), "rb").read()).hexdigest() != want:
                return None
        return prof
    except (OSError, KeyError, ValueError, TypeError):
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--gaussians", type=int, default=500_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-api-path", action="store_true", help="skip timing the reference-shaped eager route")
    ap.add_argument("--cpu-sample-div", type=int, default=4, help="CPU sample = config / div^2 (area and N)")
    ap.add_argument("--sync-m", action="store_true", help="read the intersection count back every step")
    ap.add_argument("--graph", dest="graph", action="store_true", default=None,
                    help="replay the step from a captured hipGraph (default at 1 GPU)")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="eager dispatch")
    ap.add_argument("--dp-plain", action="store_true",
                    help="N > 1: all-reduce the whole flat gradient (118 MB at config B) instead of the compact exchange "
                         "(geometry all-reduce + all-gather of the per-view colour gradients, ~2.6x fewer bytes)")
    ap.add_argument("--plain-adam", action="store_true",
                    help="write the 48 N SH-coefficient gradients in the projection backward and read them in the plain "
                         "fused Adam step, instead of expanding them inside the optimiser pass (qed_adam_step_sh)")
    ap.add_argument("--dp-one-graph", action="store_true",
                    help="N > 1: try to capture the WHOLE data-parallel step, collectives included, as one hipGraph, with the "
                         "colour-gradient all-gather issued ahead of the projection backward (also QED_BENCH_DP_ONE_GRAPH=1); "
                         "falls back to the three graphs around eager collectives when the capture fails")
    ap.add_argument("--launch-timeout", type=float, default=900.0,
                    help="--gpus N > 1 without a launcher: wall limit in seconds for the ranks this process starts")
    ap.add_argument("--graph-split", action="store_true",
                    help="capture forward+backward and the Adam step as two graphs with the gradient all-reduce "
                         "between them (the default for N > 1; this flag forces it at N = 1 for testing)")
    return ap.parse_args()


def make_scene(n, w, h, rank, dev):
    """SURVEY 8(d) generator (seed 1234 + config index 1); identical Gaussians on every rank,
    camera k yawed by 5k degrees."""
    from qed_splatter_amd.scene import synthetic_scene
    sc = synthetic_scene(n, w, h, seed=1235, n_cameras=max(rank + 1, 1))
    sc["camera_to_worlds"] = sc["camera_to_worlds"][rank:rank + 1]
    sc["Ks"] = sc["Ks"][rank:rank + 1]
    return {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sc.items()}


def cpu_baseline(args, m_full=None):
    """The CPU oracle (a port: the reference has no CPU rasterizer) timed on this box's
    host cores on a bounded sample: config B shrunk by div^2 in area AND Gaussian count (same
    Gaussians-per-pixel density), fp32, forward + loss + backward."""
    from oracle import splat_oracle as O           # cpu_baseline leg only
    div = args.cpu_sample_div
    n, w, h = args.gaussians // (div * div), args.width // div, args.height // div
    # the GPU box gives one GPU a 16-core share of the host: more threads than that only thrash
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    sc = O.synthetic_scene(n, w, h, seed=1235)
    names = ("means", "scales", "quats", "opacities", "features_dc", "features_rest")

    def one():
        ps = {k: sc[k].clone().requires_grad_(True) for k in names}
        out = O.splatfacto_outputs(ps["means"], ps["scales"], ps["quats"], ps["opacities"], ps["features_dc"],
                                   ps["features_rest"], sc["camera_to_worlds"], sc["Ks"], w, h, sc["background"])
        loss = O.main_loss(out["rgb"], sc["gt_rgb"], 0.2) + O.depth_l1_loss(out["depth"], sc["gt_depth"])
        loss.backward()
        return out["info"]["flatten_ids"].numel()

    import gc
    import resource

    def rss_gb():                                    # current resident set (not the process's historical peak)
        try:
            with open("/proc/self/statm") as f:
                return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 2 ** 30
        except (OSError, ValueError, IndexError):
            return 0.0

    def mem_available_gb():
        try:
            with open("/proc/meminfo") as f:
                for ln in f:
                    if ln.startswith("MemAvailable:"):
                        return int(ln.split()[1]) / 2 ** 20
        except OSError:
            pass
        return 0.0

    import threading
    gc.collect()
    rss0 = rss_gb()
    seen = [rss0]
    stop = threading.Event()

    def watch():                                     # resident set sampled every 10 ms while the first iteration runs
        while not stop.wait(0.01):
            seen[0] = max(seen[0], rss_gb())
    th = threading.Thread(target=watch, daemon=True)
    th.start()
    t0 = time.perf_counter()
    m = one()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    # what ONE sample iteration holds at its peak (the autograd graph of every tile's blend, alive until backward)
    peak_sample = max(seen[0] - rss0, 0.0)
    iters = 1
    while dt < 12.0 and iters < 16:                 # about 10-15 s of CPU work on the GPU box's host share
        one()
        iters += 1
        dt = time.perf_counter() - t0
    sample_it_s = iters / dt
    # ONE iteration of the configuration itself ("config B timed for 1 fwd+bwd iteration (minutes) or, if > 10
    # min, extrapolated").  The oracle keeps every tile's [pixels x run] intermediates alive until backward: the
    # full size needs div^2 times the sample's peak.  It runs only where that fits the host with room to spare and the
    # sample says it takes under five minutes; otherwise the reason is recorded with the measured numbers.
    full = {"workload": f"{args.gaussians} Gaussians @ {args.width}x{args.height} (the configuration itself), fp32"}
    # memory and time of the oracle go with pixels-per-tile x list length, i.e. with the intersection count M (measured:
    # ~60 B x 256 x M resident at the peak): the full configuration has div^2 times the tiles AND longer runs per tile
    # (splat radii scale with the image), so the factor is M_full / M_sample -- from the GPU run's own count of the
    # reference's list when given, else the pessimistic div^4
    scale = (m_full / max(m, 1)) if m_full else float(div) ** 4
    est_gb = peak_sample * scale
    est_s = (dt / iters) * scale
    avail_gb = mem_available_gb()
    try:                                             # a container's own limit, where there is one
        with open("/sys/fs/cgroup/memory.max") as f:
            v = f.read().strip()
            if v.isdigit():
                avail_gb = min(avail_gb, int(v) / 2 ** 30 - rss_gb())
    except OSError:
        pass
    limit_gb = float(os.environ.get("QED_BENCH_CPU_FULL_MAX_GB", "150"))
    if div == 1:
        full.update(skipped="the sample is the full configuration")
    elif peak_sample <= 0.0:
        full.update(skipped="peak memory of the sample could not be measured")
    elif 1.3 * est_gb > limit_gb or 1.3 * est_gb > 0.6 * avail_gb or est_s > 300.0:
        full.update(skipped=f"estimated from the sample: {est_gb:.0f} GiB peak resident memory (the oracle's per-tile autograd "
                            f"graphs; {avail_gb:.0f} GiB available, limit {limit_gb:.0f}) and {est_s:.0f} s")
    else:
        del sc
        gc.collect()
        nf, wf, hf = args.gaussians, args.width, args.height
        sf = O.synthetic_scene(nf, wf, hf, seed=1235)
        tf = time.perf_counter()
        psf = {k: sf[k].clone().requires_grad_(True) for k in names}
        outf = O.splatfacto_outputs(psf["means"], psf["scales"], psf["quats"], psf["opacities"], psf["features_dc"],
                                    psf["features_rest"], sf["camera_to_worlds"], sf["Ks"], wf, hf, sf["background"])
        (O.main_loss(outf["rgb"], sf["gt_rgb"], 0.2) + O.depth_l1_loss(outf["depth"], sf["gt_depth"])).backward()
        dtf = time.perf_counter() - tf
        full.update(iters=1, seconds=round(dtf, 2), iters_per_s=1.0 / dtf, intersections=outf["info"]["flatten_ids"].numel(),
                    peak_rss_gib=round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20, 1),
                    estimated_from_sample_gib=round(est_gb, 1))
        del sf, psf, outf
        gc.collect()
    # BASELINE.json configs[0] -- 10 k Gaussians, one camera @ 256 x 256, the plumbing case -- timed IN FULL ("config
    # A timed in full (fwd+bwd, 10 iterations)"): the only configuration the CPU runs at its real size
    na, wa, ha = 10_000, 256, 256
    sa = O.synthetic_scene(na, wa, ha, seed=1234)

    def one_a():
        ps = {k: sa[k].clone().requires_grad_(True) for k in names}
        out = O.splatfacto_outputs(ps["means"], ps["scales"], ps["quats"], ps["opacities"], ps["features_dc"],
                                   ps["features_rest"], sa["camera_to_worlds"], sa["Ks"], wa, ha, sa["background"])
        (O.main_loss(out["rgb"], sa["gt_rgb"], 0.2) + O.depth_l1_loss(out["depth"], sa["gt_depth"])).backward()
        return out["info"]["flatten_ids"].numel()

    ma = one_a()                                    # (untimed: first-call allocations)
    ta = time.perf_counter()
    for _ in range(10):
        one_a()
    dta = time.perf_counter() - ta
    config_a = {"workload": f"{na} Gaussians, 1 cam @ {wa}x{ha} (BASELINE configs[0]), fp32, fwd+loss+bwd, full size",
                "iters": 10, "seconds": round(dta, 3), "iters_per_s": 10 / dta, "intersections": ma}
    ran_full = "iters_per_s" in full
    return {
        # the configuration's own iteration when it ran; otherwise the sample's rate / div^2
        "value": full["iters_per_s"] if ran_full else sample_it_s / (div * div),
        "unit": "train iters/s" + ("" if ran_full else " (full-workload equivalent)"),
        "cores": cores, "kind": "port", "config_a": config_a, "full_iteration": full,
        "sample": (f"ONE iteration of the full configuration in {full['seconds']} s (peak {full['peak_rss_gib']} GiB); beside it " if ran_full
                   else f"full configuration not run ({full.get('skipped')}); ") +
                  f"{n} Gaussians @ {w}x{h} (config / {div * div}, same density), fp32, fwd+loss+bwd, "
                  f"{iters} iters in {dt:.1f}s = {sample_it_s:.3f} sample-iters/s, M={m}"
                  + ("" if ran_full else f"; value = that / {div * div}"),
    }


def api_path_ms(args, sc, dev, optimizer: str, separate_params: bool = False, host: dict = None,
                graph_segments="always") -> float:
    """ms per training step through the REFERENCE's own call sequence, eager dispatch, same scene and same steps
    (W warm-up + K timed from the initial parameters) as the headline number:

        zero_grad -> get_outputs(camera) -> get_metrics_dict -> get_loss_dict -> sum(loss_dict) -> backward ->
        one optimiser per parameter group (config.py:44-68) stepped in turn -> the means' ExponentialDecay scheduler

    (/root/reference/qed_splatter/model.py:199-321, 120-197, 73-118 driven by Nerfstudio's Trainer.train_iteration).
    ``optimizer``: "qed" = QedAdam (a torch.optim.Optimizer subclass AdamOptimizerConfig._target can name; the six
    instances share one fused launch), "torch" = six torch.optim.Adam exactly as the reference configures them.
    ``separate_params``: the six Parameters as SEPARATE tensors, which is how Nerfstudio's parent class holds them
    (model.py:12,50-58) -- QedAdam then keeps torch.optim.Adam's per-parameter state and launches once per group."""
    import functools
    from qed_splatter_amd.model import (FlatAdam, PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig, QedAdam,
                                        exponential_decay_lr)
    n, w, h = args.gaussians, args.width, args.height
    cfg = QEDSplatterModelConfig.synthetic(sh_degree=3, sh_degree_interval=1, graph_segments=graph_segments)
    model = QEDSplatterModel(cfg, separate_params=separate_params,
                             **{k: sc[k].clone() for k in ("means", "scales", "quats", "opacities", "features_dc",
                                                           "features_rest")})
    model.step = 30000
    model.train()
    K = sc["Ks"][0].cpu()
    # (Nerfstudio's datamanager hands the camera index along: the model keeps the compositing forward's launch order per camera)
    cam = PinholeCameras(sc["camera_to_worlds"], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), w, h,
                         metadata={"cam_idx": 0})
    batch = {"image": sc["gt_rgb"].contiguous(), "depth_image": sc["gt_depth"].contiguous()}
    cls = QedAdam if optimizer == "qed" else torch.optim.Adam
    lrs = FlatAdam.DEFAULT_LRS                                   # = config.py:44-68
    opts = {k: cls([model.gauss_params[k]], lr=lrs[k], eps=1e-15) for k in
            ("means", "features_dc", "features_rest", "opacities", "scales", "quats")}        # config.py's order
    lr_final, max_steps = FlatAdam.MEANS_SCHEDULE
    sched = torch.optim.lr_scheduler.LambdaLR(
        opts["means"], lambda s: exponential_decay_lr(s, lrs["means"], lr_final, max_steps) / lrs["means"])

    def step():
        for o in opts.values():
            o.zero_grad(set_to_none=True)
        outputs = model.get_outputs(cam)
        metrics = model.get_metrics_dict(outputs, batch)
        loss_dict = model.get_loss_dict(outputs, batch, metrics)
        functools.reduce(torch.add, loss_dict.values()).backward()
        for o in opts.values():
            o.step()
        sched.step()

    import gc
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # This route is bound by what the host spends per step, so a full collection of Python's cyclic garbage collector
    # landing in the timed region (70-150 ms over a heap that holds all of torch: seen as 2.6-3.8 ms/step in one of three
    # runs of one process) says nothing about the route: collect now, keep the collector off for the timed steps.
    gc.collect()
    gc.disable()
    from qed_splatter_amd.rasterization import _workspace
    ws = _workspace(dev)
    try:
        waited0 = ws.waited_s
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        t_issue = time.perf_counter() - t0             # host time to enqueue the steps (the GPU may still be running)
        waited = ws.waited_s - waited0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        gc.enable()
    if host is not None:
        # enqueue = wall time until the last step was enqueued.  The host may run at most ONE frame ahead of the device (each
        # get_outputs first looks at the previous frame's intersection count), so when the device is the slower side this
        # is about the step time whatever the host costs; busy = enqueue minus the time spent waiting for that count: what
        # the host really spends per step, and what bounds the step on a slower host
        host["enqueue_ms_per_step"] = t_issue / args.steps * 1e3
        host["busy_ms_per_step"] = (t_issue - waited) / args.steps * 1e3
        host["wait_frac"] = waited / max(t_issue, 1e-12)
        host["graph_segments"] = bool(model.__dict__.get("_segments") and model._segments.segments)
    del model, opts
    torch.cuda.empty_cache()
    return dt / args.steps * 1e3


_JSON_FD = 1


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as FRESH child processes
    (one per GPU, torch.distributed.run over 127.0.0.1) before this process has made any GPU call, relay their
    output (rank 0 prints the JSON line) and return their exit status.  Nothing is re-exec'd.

    A wedged rank must not sit until the driver's own limit: the launcher and its ranks run in a process group of their
    own, and after ``--launch-timeout`` seconds (QED_BENCH_LAUNCH_TIMEOUT; default 900) that group is terminated (then
    killed) and this process exits non-zero.  The ranks themselves give every collective / rendezvous 120 s
    (init_process_group(timeout=...))."""
    import signal
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    limit = float(os.environ.get("QED_BENCH_LAUNCH_TIMEOUT", args.launch_timeout))
    print(f"[bench] launching {args.gpus} ranks (wall limit {limit:.0f} s): {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, start_new_session=True)     # its own process group: exactly what we started
    try:
        return proc.wait(timeout=limit)
    except subprocess.TimeoutExpired:
        print(f"[bench] the ranks did not finish within {limit:.0f} s: terminating process group {proc.pid}",
              file=sys.stderr, flush=True)
    except KeyboardInterrupt:
        print("[bench] interrupted: terminating the ranks", file=sys.stderr, flush=True)
    for sig, grace in ((signal.SIGTERM, 15.0), (signal.SIGKILL, 5.0)):
        try:
            os.killpg(proc.pid, sig)
        except ProcessLookupError:
            break
        try:
            proc.wait(timeout=grace)
            break
        except subprocess.TimeoutExpired:
            continue
    return 124


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))                 # before torch.cuda is touched in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries ONE line, rank 0's JSON.  Libraries print there too (RCCL writes a version banner to stdout when its
    # first communicator comes up): everything else of this process goes to stderr, the line through the saved descriptor
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    # rehearsal aid for a one-GPU box: QED_BENCH_REHEARSE=1 puts every rank on cuda:0 and uses gloo, which
    # exercises the multi-rank control flow (split graphs, rank agreement, barriers) without RCCL
    rehearse = os.environ.get("QED_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # QED_BENCH_RCCL_SELF=1 (one rank only): the N > 1 path -- split graphs, the compact exchange overlapped with the
    # optimiser, barriers, rank agreement -- through the real `nccl` (= RCCL) backend in a process group of ONE rank: what
    # a one-GPU box can drive of it (the collectives are issued and waited for; there is no peer, so nothing is measured)
    rccl_self = world == 1 and os.environ.get("QED_BENCH_RCCL_SELF") == "1"
    if rccl_self:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            free_port = s.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    multi = world > 1 or rccl_self
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from datetime import timedelta
        # a rank that never arrives must fail the others within two minutes, not hold them until the driver's limit
        limit = timedelta(seconds=float(os.environ.get("QED_BENCH_COLLECTIVE_TIMEOUT", "120")))
        if rehearse:
            dist.init_process_group("gloo", timeout=limit)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=limit)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    # Work on an explicit (non-legacy) stream from the start: autograd pins each leaf's gradient
    # accumulation to the stream it first ran on, and nothing may touch the legacy default stream
    # while the step is being captured into a hipGraph.
    torch.cuda.set_stream(torch.cuda.Stream(device=dev))

    from qed_splatter_amd import _lib as L
    from qed_splatter_amd.model import FlatAdam, PinholeCameras, QEDSplatterModel, QEDSplatterModelConfig
    from qed_splatter_amd import parallel as P
    from qed_splatter_amd.parallel import allreduce_flat_grad, exchange_grads_compact_begin
    P.FORCE_COLLECTIVES = rccl_self
    L.load()

    n, w, h = args.gaussians, args.width, args.height
    sc = make_scene(n, w, h, rank, dev)
    cfg = QEDSplatterModelConfig.synthetic(sh_degree=3, sh_degree_interval=1)
    model = QEDSplatterModel(cfg, **{k: sc[k] for k in ("means", "scales", "quats", "opacities", "features_dc",
                                                         "features_rest")})
    model.step = 30000                       # full SH degree
    K = sc["Ks"][0].cpu()
    cam = PinholeCameras(sc["camera_to_worlds"], float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]), w, h)
    batch = {"image": sc["gt_rgb"].contiguous(), "depth_image": sc["gt_depth"].contiguous()}
    bg = torch.zeros(3, device=dev)
    opt = FlatAdam(model, means_schedule=FlatAdam.MEANS_SCHEDULE)     # the reference's optimiser config (config.py:44-68)

    # SH-coefficient gradients as (3 colour gradients + view) per Gaussian, expanded inside the optimiser pass
    # (qed_adam_step_sh): the default; --dp-plain / --plain-adam materialise the 48 N gradients instead
    # Training changes the scene (and with it the list length M and the step time: random ground truth drives
    # opacities down), so every timed region measures the SAME steps of the same run: parameters and optimiser
    # state are put back to the initial scene, W warm-up steps are taken, then K steps are timed.
    init_params = model.flat_params.detach().clone()

    def restore():
        with torch.no_grad():
            model.flat_params.copy_(init_params)
            opt.exp_avg.zero_()
            opt.exp_avg_sq.zero_()
            opt.dev_state.zero_()
        opt.t = 0
        opt.set_lr("means", opt._means_lr_init)

    dp_compact = not args.dp_plain and not args.plain_adam
    # every rank renders ONE fixed camera step after step: the compositing forward takes its launch order from the previous
    # frame of that camera (model.fused_loss, frame_key); QED_BENCH_FRAME_KEY=0 is the A/B switch
    FRAME_KEY = None if os.environ.get("QED_BENCH_FRAME_KEY", "1") == "0" else 0
    fused_sh = dp_compact
    # N > 1: row capacity of the sparse colour-gradient message, agreed by all ranks after the first step (None: the dense
    # message is no larger -- SURVEY 8d's scene is 95 % visible; QED_BENCH_DP_SPARSE=1 forces the sparse form for rehearsals)
    sparse_cap = [None]

    def exchange_and_step(adam_sh, adam_leading, adam_all, in_graph=False):
        """N > 1.  Compact exchange: the all-gather of the colour gradients is issued first and the SH part of the
        optimiser (two thirds of its time) runs behind it while the geometry all-reduce is still on the links; the
        leading groups follow that all-reduce.  --dp-plain: one all-reduce of the flat gradient, then the step."""
        if dp_compact:
            # in_graph: the message was assembled at the end of the forward+backward graph and the SH graph folds the
            # gathered overflow words itself (four eager launches less per step)
            if sparse_cap[0] is not None:
                # the sparse colour-gradient message (rows of the Gaussians this rank saw): packs and unpacks eagerly
                ex = exchange_grads_compact_begin(model, world, sparse_cap=sparse_cap[0])
                ex.wait_views()
                if in_graph:
                    P.fold_skip_words(model)        # (the SH graph's own fold then repeats it: idempotent)
            else:
                ex = exchange_grads_compact_begin(model, world, prepared=in_graph, fold=not in_graph)
                ex.wait_views()
            adam_sh()
            ex.wait_geometry()
            adam_leading()
        else:
            allreduce_flat_grad(model, world)
            adam_all()

    def step(sync):
        for p in model.parameters():
            p.grad = None
        losses = model.fused_loss(cam, batch, background=bg, sync=sync, compact_sh_grad=dp_compact, frame_key=FRAME_KEY)
        model.backward_fused(losses)
        if multi:
            exchange_and_step(lambda: opt.step(fused_sh=True, part=1), lambda: opt.step(fused_sh=True, part=2),
                              lambda: opt.step(fused_sh=fused_sh))
        else:
            opt.step(fused_sh=fused_sh)
        return losses

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    # The unit count of SURVEY 8(d) is the length of the REFERENCE's list (gsplat's 3-sigma squares); the tight
    # lists this run uses are a subset that renders the same image.  One un-timed forward gives that length
    # (then the intersection-buffer capacity is re-calibrated for the lists actually used).
    M_ref = None
    if cfg.tight_tile_lists:
        from qed_splatter_amd.rasterization import _workspace
        cfg.tight_tile_lists = False
        with torch.no_grad():
            model.fused_loss(cam, batch, background=bg, sync=True)
        M_ref = int(model.info["n_isects"])
        cfg.tight_tile_lists = True
        _workspace(dev).reset()
    # first step is synchronous: it calibrates the intersection-buffer capacity and gives M
    step(True)
    M = int(model.info["n_isects"])
    M_ref = M if M_ref is None else M_ref
    n_vis = int((model.info["radii"] > 0).sum())
    log(f"scene ready: N={n} visible={n_vis} M={M}")
    if multi and dp_compact:
        sparse_cap[0] = P.sparse_message_capacity(model, world)          # (one MAX all-reduce; None: dense is no larger)
        if sparse_cap[0] is None and os.environ.get("QED_BENCH_DP_SPARSE") == "1":
            sparse_cap[0] = (n + 3) // 4 * 4
        log("colour-gradient message: " + ("dense (3 N floats per rank)" if sparse_cap[0] is None else
                                           f"sparse, {sparse_cap[0]} rows of {n} per rank"))
    restore()
    for _ in range(args.warmup):
        step(args.sync_m)
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    L.TIMER.reset()
    L.TIMER.active = (rank == 0)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(args.sync_m)
    barrier()
    dt = time.perf_counter() - t0
    L.TIMER.active = False
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    kern = L.TIMER.summary()
    log(f"instrumented region: {dt / args.steps * 1e3:.3f} ms/step")

    # a second, un-instrumented timed region gives the headline number (event records cost host time).
    # Default at N = 1: the whole step replayed from one captured hipGraph (same kernels, same work;
    # only the host dispatch cost is removed).  --no-graph times eager dispatch instead.
    # N > 1: forward+backward and the Adam step are captured as TWO graphs and the RCCL all-reduce of the
    # flat gradient is issued between them (a collective is not captured: it stays an ordinary call on
    # the process group's stream).
    use_graph = args.graph if args.graph is not None else True
    split = use_graph and (multi or args.graph_split)
    run = lambda: step(args.sync_m)
    dispatch, graphs = "eager", []
    if use_graph:
        from qed_splatter_amd.graph import GraphedTrainStep

        def fwd_bwd(tick_for=None):
            for p in model.parameters():
                p.grad = None
            losses = model.fused_loss(cam, batch, background=bg, sync=False, compact_sh_grad=dp_compact,
                                      optimizer=tick_for, frame_key=FRAME_KEY)
            model.backward_fused(losses)
            return losses

        def adam_only():
            opt.step(device_state=True, fused_sh=fused_sh)
            return {}

        def adam_sh_part():
            opt.step(device_state=True, fused_sh=True, part=1)
            return {}

        def fwd_bwd_message():
            losses = fwd_bwd()
            P.prepare_compact_message(model, world)
            return losses

        def fold_adam_sh_part():
            P.fold_skip_words(model)
            return adam_sh_part()

        def adam_leading_part():
            opt.step(device_state=True, fused_sh=True, part=2)
            return {}

        def graph_step():
            # one graph for the whole step: the loss pass's fold launch also advances the optimiser's device step
            # state (FlatAdam.take_tick), so the Adam launches that follow need no one-thread launch for it
            losses = fwd_bwd(opt if fused_sh else None)
            adam_only()
            return losses

        # N > 1: ONE graph with the collectives captured is the default (0.987 against 1.028 ms in a group of one,
        # profiles/r04_bench_rccl_self.json); QED_BENCH_DP_ONE_GRAPH=0 keeps the three graphs around eager collectives, =1
        # issues the all-gather ahead of the projection backward (--dp-one-graph), =2 / unset behind the whole backward.
        # Whatever is asked for, the three-graph form is what runs unless EVERY rank captured the one-graph form.
        # (gloo rehearsals stage through the host: nothing to capture there)
        one_graph_env = os.environ.get("QED_BENCH_DP_ONE_GRAPH", "1" if args.dp_one_graph else ("0" if rehearse else "2"))
        one_graph = [one_graph_env in ("1", "2")]

        def capture_all():
            """(run, dispatch, graphs): every graph of the step captured afresh -- also after an intersection overflow, when
            the forward+backward graph gets new (larger) buffers and with them new static .grad tensors the optimiser
            graphs must read."""
            opt.dev_state[0] = float(opt.t)           # hand the step counter over to the device-side state
            if split and multi and dp_compact and one_graph[0]:
                # the whole data-parallel step in ONE graph: PyTorch captures RCCL collectives issued on (or joined to) the
                # capture stream; the all-gather leaves from inside the backward pass, ahead of the projection backward
                def dp_step():
                    for p in model.parameters():
                        p.grad = None
                    losses = model.fused_loss(cam, batch, background=bg, sync=False, compact_sh_grad=True, optimizer=opt,
                                              frame_key=FRAME_KEY)
                    if one_graph_env == "1":                 # the gather ahead of the projection backward
                        P.backward_with_early_gather(model, losses, world)
                    else:
                        model.backward_fused(losses)
                    ex = exchange_grads_compact_begin(model, world)
                    ex.wait_views()
                    adam_sh_part()
                    ex.wait_geometry()
                    adam_leading_part()
                    return losses
                g1 = None
                try:
                    g1 = GraphedTrainStep(dp_step, dev, warmup=3, check_every=0)
                except Exception as e:
                    log(f"one-graph capture of the data-parallel step failed ({type(e).__name__}: {e}); three graphs instead")
                    torch.cuda.synchronize()
                    opt.drop_tick()
                # the ranks must run the SAME form (a rank replaying captured collectives against peers that issue eager
                # ones in another order would hang): one graph only if every rank has it
                agreed = torch.tensor([1 if g1 is not None else 0], device=dev, dtype=torch.int32)
                dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
                if int(agreed):
                    where = "ahead of the projection backward" if one_graph_env == "1" else "behind the backward pass"
                    return g1.replay, (f"ONE hipGraph with the collectives inside: all-gather {where}, SH groups behind it "
                                       "while the geometry all-reduce is on the links, leading groups"), [g1]
                if g1 is not None:
                    log("another rank could not capture the one-graph form; three graphs instead")
                    del g1
                    opt.drop_tick()
                one_graph[0] = False
            if split:
                g_fb = GraphedTrainStep(fwd_bwd_message if multi and dp_compact else fwd_bwd, dev, warmup=3, check_every=0)
                g_fb.replay()                          # fills the captured (static) .grad buffers before Adam's
                                                       # warm-up run reads them; keep the replicas identical
                if multi and dp_compact:
                    ex = exchange_grads_compact_begin(model, world, prepared=True)
                    ex.wait_views()
                    ex.wait_geometry()
                    g_sh = GraphedTrainStep(fold_adam_sh_part, dev, warmup=1, check_every=0)
                    g_lead = GraphedTrainStep(adam_leading_part, dev, warmup=1, check_every=0)

                    def run_():
                        g_fb.replay()
                        exchange_and_step(g_sh.replay, g_lead.replay, None, in_graph=True)
                    return run_, ("three hipGraphs (fwd+bwd | Adam SH groups | Adam leading groups): all-gather, SH groups "
                                  "behind it while the geometry all-reduce is on the links, leading groups"), [g_fb, g_sh, g_lead]
                if multi:
                    allreduce_flat_grad(model, world)
                g_adam = GraphedTrainStep(adam_only, dev, warmup=1, check_every=0)

                def run_():
                    g_fb.replay()
                    if multi:
                        allreduce_flat_grad(model, world)
                    g_adam.replay()
                return run_, "two hipGraphs (fwd+bwd | Adam) around the gradient all-reduce", [g_fb, g_adam]
            g = GraphedTrainStep(graph_step, dev, warmup=3, check_every=0)
            return g.replay, "hipGraph replay of the whole step", [g]

        try:
            run, dispatch, graphs = capture_all()
            captured = True
        except Exception as e:                         # capture is an optimisation of dispatch only
            log(f"graph capture failed ({type(e).__name__}: {e})")
            torch.cuda.synchronize()
            opt.drop_tick()
            captured = False
        if dist is not None:                           # every rank must issue the same collectives from here on
            ok = torch.tensor([1 if captured else 0], device=dev, dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            captured = bool(int(ok))
        if captured:
            log(f"step captured: {dispatch}")
        else:
            use_graph, dispatch = False, "eager (graph capture failed)"
            run = lambda: step(args.sync_m)

    from qed_splatter_amd.rasterization import _workspace as _ws_of
    ws = _ws_of(dev)
    overflows_seen = ws.overflows

    def replays_had_room() -> bool:
        """True when no replay since the last call overflowed the captured intersection buffer ON ANY RANK.  A replay that
        overflows renders an empty frame and its optimiser launches are no-ops on the device -- a FASTER step: a region that
        contains one is not a measurement.  Otherwise: every graph is captured again with room (on every rank: the
        decision is collective, the ranks must keep issuing the same collectives) and the caller times the region again."""
        nonlocal run, dispatch, graphs, overflows_seen
        if use_graph:
            graphs[0].check()                          # (reads the overflow word; grows the capacity and re-captures itself)
        else:
            torch.cuda.synchronize()
            ws.poll_pending()
        had = int(ws.overflows != overflows_seen)
        overflows_seen = ws.overflows
        if dist is not None:
            t = torch.tensor([had], device=dev, dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            had = int(t)
        if had and use_graph:
            log("an intersection overflow during the replays: all graphs captured again, region repeated")
            run, dispatch, graphs = capture_all()
        return not had

    def rewind():
        """Back to the initial scene + W warm-up steps: what precedes every timed region."""
        restore()
        for _ in range(args.warmup):
            run()
        for _ in range(3):
            if replays_had_room():
                return
            restore()
            for _ in range(args.warmup):
                run()
        raise SystemExit("bench: intersection overflows persisted through three re-captures")

    overflows_before = ws.overflows
    for attempt in range(3):
        rewind()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run()
        t_issue = time.perf_counter() - t0               # host time to enqueue the steps (logged, not reported)
        barrier()
        dt2 = time.perf_counter() - t0
        log(f"host enqueue time {t_issue / args.steps * 1e3:.3f} ms/step")
        # a region in which a replay overflowed trained on empty frames with the optimiser skipped: not a measurement
        if replays_had_room():
            break
    else:
        raise SystemExit("bench: every attempt at the timed region contained an intersection overflow")
    if dist is not None:
        t = torch.tensor([dt2], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt2 = float(t)

    # per-step spread (median and p10/p90): a third, short region with a HIP event between
    # consecutive steps -- kept out of the headline region, which contains nothing but the steps
    pct = None
    if world == 1:
        n_ev = min(max(args.steps, 20), 100)
        rewind()
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_ev + 1)]
        evs[0].record()
        for i in range(n_ev):
            run()
            evs[i + 1].record()
        torch.cuda.synchronize()
        per = sorted(a.elapsed_time(b) for a, b in zip(evs[:-1], evs[1:]))
        pct = {"p10": round(per[int(0.1 * (n_ev - 1))], 4), "p50": round(per[(n_ev - 1) // 2], 4),
               "p90": round(per[int(0.9 * (n_ev - 1))], 4), "steps": n_ev}

    # list length after W + K steps of training (the scene drifts; config.intersections is the first step's)
    rewind()
    for _ in range(args.steps):
        run()
    with torch.no_grad():
        model.fused_loss(cam, batch, background=bg, sync=True)
    M_end = int(model.info["n_isects"])

    if rank == 0:
        ms_step = dt2 / args.steps * 1e3
        n_px = w * h
        # dominant kernel = compositing backward; algorithmic bytes: 92 B per list entry the launch
        # processed + 28 B per pixel.  The list shrinks as training proceeds (config.intersections ->
        # intersections_after_timed_steps), so the mean of the two ends of the run is used.
        dom = "qed_composite_bwd"
        dom_ms = kern.get(dom, (0, float("nan")))[1]
        M_proc = 0.5 * (M + M_end)
        alg_bytes = 92.0 * M_proc + 28.0 * n_px
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        # Counter-derived figures of that kernel come from COMMITTED rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
        # cannot share a pass and counters cannot be read from inside this process): valid only for this workload and
        # for the kernel version named in the profile; null otherwise.
        traffic = valu_frac = prof_note = None
        if (n, w, h) == (500_000, 1920, 1080):
            prof, vp = committed_counters(PMC_TRAFFIC), committed_counters(PMC_VALU)
            try:
                if prof is not None:
                    kk = prof["kernels"]["qed::composite_bwd_kernel<4>"]
                    traffic = (2.0 * kk["fetch_size_kb"] + kk["write_size_kb"]) * 1024.0    # gfx950: FETCH_SIZE counts half
                if vp is not None:
                    valu_frac = vp["kernels"]["qed::composite_bwd_kernel<4>"]["valu_issue_frac"]
                if prof is not None or vp is not None:
                    prof_note = f"from committed profiles {PMC_TRAFFIC} / {PMC_VALU} (kernels " \
                                f"{(prof or vp).get('kernels_version', '?')}; source hashes match this tree), not measured in this run"
                else:
                    prof_note = f"{PMC_TRAFFIC} / {PMC_VALU} were taken on other kernel sources (sha256 mismatch): not quoted"
            except (KeyError, TypeError):
                pass
        roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": prof_note,
                "algorithmic_bytes_per_launch": alg_bytes, "list_entries_per_launch": M_proc,
                "list_entries_reference": M_ref, "kernel_ms": dom_ms,
                "kernel_ms_covers": "HIP events on the launch stream around the qed_composite_bwd ENTRY POINT, eager: the "
                                    "one-workgroup ordering launch (tile_order_kernel, ~10 us) + composite_bwd_kernel + the two "
                                    "event records; the rocprofv3 kernel trace of the same command lists the two kernels "
                                    "separately (profiles/r05_bench_kernel_stats_v3.txt: 9.8 + 276.1 us on that box)",
                # the same launch priced on the list the REFERENCE operator would have walked for this image (SURVEY 8d's
                # unit is gsplat's intersection): a shorter list for the same image lowers `frac` while the kernel gets
                # faster, so both are given; `achieved` / `frac` stay on what the launch really processed
                "frac_at_reference_list": (92.0 * M_ref * (M_proc / max(M, 1)) + 28.0 * n_px) / (dom_ms * 1e-3) / 1e9 / 8000.0,
                "note": "algorithmic bytes = 92 B x (list entries this launch processed) + 28 B x pixels (SURVEY 8d).  The "
                        "reference (gsplat) would list list_entries_reference entries for the same image; this run lists "
                        "only the tiles in which some pixel can reach alpha >= 1/255 (exact per-tile test).  The kernel is "
                        "not HBM bound (see roofline_issue): measured traffic is BELOW the algorithmic bytes because culled "
                        "/ early-terminated entries are never gathered."}
        roof_issue = {"bound": "valu_issue", "kernel": dom, "frac": valu_frac, "source": prof_note,
                      "definition": "SQ_ACTIVE_INST_VALU / (SIMDs x GRBM_GUI_ACTIVE / 8 / 4) quad-cycles; a wave64 "
                                    "cross-lane op (v_readlane 11.5, v_permlane*_swap 8.4, DPP 4.3 cycles) occupies the "
                                    "pipe several times longer than a plain op (2.2-2.5): DESIGN.md section 4"}
        out = {
            "metric": "train iters/sec @ 1080p, 500k Gaussians (fwd + loss + bwd + Adam; camera-steps/s over all GPUs)",
            "value": world * args.steps / dt2, "unit": "iters/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{n} Gaussians, SH deg 3, {world} cam(s) @ {w}x{h}, 1 per GPU, fwd+bwd with "
                                   f"depth-L1 + (0.8 L1 + 0.2 (1-SSIM)) RGB loss + fused Adam",
                       "gaussians": n, "visible": n_vis, "intersections": M, "intersections_after_timed_steps": M_end,
                       "intersections_reference_list": M_ref,
                       "forward_launch_order": "costliest tiles first, from the work counts of the PREVIOUS frame of the same "
                                               "camera (frame_key; a scheduling hint: images and gradients unchanged; "
                                               "QED_BENCH_FRAME_KEY=0 turns it off: raster order + quadrant tail)"
                                               if FRAME_KEY is not None else "raster order + a tail of quadrant waves",
                       "tile_lists": "exact (the tiles of the 3-sigma square in which some pixel can reach alpha >= 1/255; "
                                     "images and gradients identical)" if cfg.tight_tile_lists else "gsplat 3-sigma squares",
                       "width": w, "height": h,
                       "parallelism": (f"dp{world} (camera-sharded; " + ("geometry all-reduce + all-gather of per-view colour gradients"
                                                                    if dp_compact else "flat-gradient all-reduce") + ")")
                       if world > 1 else ("single (QED_BENCH_RCCL_SELF: the N > 1 exchange path through RCCL in a group of one rank)"
                                          if rccl_self else "single"),
                       "async_intersection_count": not args.sync_m,
                       "dispatch": dispatch,
                       # frames that overflowed the intersection buffer anywhere in this run (a region that held one was
                       # timed again: see replays_had_room)
                       "intersection_overflows": ws.overflows - overflows_before},
            "msplats_per_s": n_vis * world / (dt2 / args.steps) / 1e6,
            # Msplats rasterized / s = N_visible / t_fwd (t_fwd = projection + binning + compositing
            # forward, HIP-event sums of the instrumented region), the list rate M / t_fwd, and the step rate
            # with the optimiser launch left out
            "forward_ms": round(sum(kern.get(k, (0, 0.0))[1] for k in
                                    ("qed_camera_setup", "qed_project_fwd", "qed_bin_tiles", "qed_composite_fwd")), 4),
            "msplats_per_s_forward": n_vis / max(sum(kern.get(k, (0, 0.0))[1] for k in
                                                     ("qed_camera_setup", "qed_project_fwd", "qed_bin_tiles",
                                                      "qed_composite_fwd")), 1e-9) / 1e3,
            "mintersections_per_s_forward": M_ref / max(sum(kern.get(k, (0, 0.0))[1] for k in
                                                            ("qed_camera_setup", "qed_project_fwd", "qed_bin_tiles",
                                                             "qed_composite_fwd")), 1e-9) / 1e3,
            "iters_per_s_without_optimizer": world * 1e3 / max(ms_step - max(kern.get("qed_adam_step", (0, 0.0))[1],
                                                                                      kern.get("qed_adam_step_sh", (0, 0.0))[1]), 1e-9),
            "ms_per_step_instrumented": dt / args.steps * 1e3,
            "ms_per_step_percentiles": pct,
            "kernels_ms": {k: round(v[1], 4) for k, v in sorted(kern.items())},
            "roofline": roof,
            "roofline_issue": roof_issue,
        }
        log(f"timed region: {ms_step:.3f} ms/step")
        if world == 1 and not args.no_api_path:
            # the reference-shaped route, driver-timed in the same run (see api_path_ms)
            # The route's get_outputs / backward can replay captured hipGraphs (segments.py).  The default policy
            # (config.graph_segments = True) captures a shape once it has been stable for ~240 calls AND the host is the
            # slower side -- longer than this benchmark runs, so both forms are timed here and the headline is the one
            # that policy settles on for this box (its rule applied to the eager run's measured wait fraction).
            from qed_splatter_amd.segments import SegmentCache
            eager_host, seg_host = {}, {}
            api_eager = api_path_ms(args, sc, dev, "qed", host=eager_host, graph_segments=False)
            api_seg = api_path_ms(args, sc, dev, "qed", host=seg_host, graph_segments="always")
            captures = eager_host["wait_frac"] <= SegmentCache.WAIT_FRAC
            api_qed, api_host = (api_seg, seg_host) if captures else (api_eager, eager_host)
            api_torch = api_path_ms(args, sc, dev, "torch", graph_segments="always" if captures else False)
            api_sep = api_path_ms(args, sc, dev, "qed", separate_params=True, graph_segments="always" if captures else False)
            out["api_path_ms_per_step"] = api_qed
            out["api_path_torch_adam_ms_per_step"] = api_torch
            out["api_path_separate_params_ms_per_step"] = api_sep
            out["api_path_host_enqueue_ms_per_step"] = api_host.get("enqueue_ms_per_step")
            out["api_path_host_busy_ms_per_step"] = api_host.get("busy_ms_per_step")
            out["api_path_graph_segments"] = (
                f"captured (the host waited for the device {100 * eager_host['wait_frac']:.0f} % of the eager run: host-bound)"
                if captures else
                f"eager (the host waited for the device {100 * eager_host['wait_frac']:.0f} % of the eager run: device-bound, "
                f"a replay buys nothing)")
            out["api_path_eager_ms_per_step"] = api_eager
            out["api_path_eager_host_busy_ms_per_step"] = eager_host.get("busy_ms_per_step")
            out["api_path_segments_ms_per_step"] = api_seg
            out["api_path_segments_host_busy_ms_per_step"] = seg_host.get("busy_ms_per_step")
            out["api_path"] = {
                "sequence": "zero_grad, get_outputs, get_metrics_dict, get_loss_dict, sum, backward, six per-group "
                            "optimisers stepped in turn, means scheduler; eager dispatch; same scene, warm-up and steps",
                "api_path_ms_per_step": "optimisers = QedAdam (torch.optim.Optimizer subclass; the six instances share "
                                        "one fused launch)",
                "api_path_torch_adam_ms_per_step": "optimisers = torch.optim.Adam as config.py:44-68 builds them",
                "api_path_separate_params_ms_per_step": "QedAdam on six SEPARATELY held Parameters (how Nerfstudio's parent "
                                                        "class keeps them): torch.optim.Adam's per-parameter state, one "
                                                        "fused launch per group",
                "api_path_host_enqueue_ms_per_step": "wall time until the last step was enqueued / steps.  The host can run at most ONE "
                                                     "frame ahead of the device (every get_outputs first looks at the previous "
                                                     "frame's intersection count), so this reads ~ the step time whenever the "
                                                     "device is the slower side, whatever the host costs",
                "api_path_host_busy_ms_per_step": "enqueue minus the time spent waiting for that count: what the host really spends "
                                                  "per step (of the form api_path_ms_per_step was timed on)",
                "api_path_eager_* / api_path_segments_*": "the route with every call eager / with get_outputs and its backward "
                                                          "replayed from captured hipGraphs behind one autograd node "
                                                          "(segments.py); api_path_graph_segments says which of the two the "
                                                          "default capture policy settles on for this box and why",
                "iters_per_s": 1e3 / api_qed}
            log(f"reference-shaped route: {api_qed:.3f} ms/step (QedAdam), {api_torch:.3f} ms/step (torch.optim.Adam), "
                f"{api_sep:.3f} ms/step (QedAdam, separate Parameters)")
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args, m_full=M_ref)
            # the same full-size configs[0] step on the GPU (fused route, eager dispatch), beside the CPU figure
            from qed_splatter_amd.scene import synthetic_scene
            sa = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in synthetic_scene(10_000, 256, 256, seed=1234).items()}
            ma = QEDSplatterModel(QEDSplatterModelConfig.synthetic(sh_degree=3, sh_degree_interval=1),
                                  **{k: sa[k] for k in ("means", "scales", "quats", "opacities", "features_dc", "features_rest")})
            ma.step = 30000
            Ka = sa["Ks"][0].cpu()
            cam_a = PinholeCameras(sa["camera_to_worlds"], float(Ka[0, 0]), float(Ka[1, 1]), float(Ka[0, 2]), float(Ka[1, 2]), 256, 256)
            batch_a = {"image": sa["gt_rgb"].contiguous(), "depth_image": sa["gt_depth"].contiguous()}

            def step_a():
                for p in ma.parameters():
                    p.grad = None
                ma.backward_fused(ma.fused_loss(cam_a, batch_a, sync=False))

            for _ in range(10):
                step_a()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(100):
                step_a()
            torch.cuda.synchronize()
            out["cpu_baseline"]["config_a"]["gpu_iters_per_s_same_workload"] = 100 / (time.perf_counter() - ta)
        os.write(_JSON_FD, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
