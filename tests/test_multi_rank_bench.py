"""The multi-rank control flow of bench.py inside the GPU test tier: `python bench.py --gpus 2`
as the driver calls it -- no launcher around it -- starts its own fresh ranks, which on a one-GPU box share cuda:0 and talk
over gloo (QED_BENCH_REHEARSE=1: RCCL needs one GPU per rank).  Everything but the transport is the N > 1 path of the
real run: per-rank cameras, split graphs around the eager collectives, the compact gradient exchange overlapped with the
optimiser, rank agreement, barriers, max-over-ranks timing, rank 0's JSON line.  No scaling figure comes out of this."""
from __future__ import annotations

import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_prints_the_contract_line(cuda):
    env = dict(os.environ, QED_BENCH_REHEARSE="1", QED_BENCH_LAUNCH_TIMEOUT="420", QED_BENCH_COLLECTIVE_TIMEOUT="120")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--gaussians", "100000", "--width", "960", "--height", "540"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]                       # rank 0 alone prints, once
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["parallelism"].startswith("dp2") and out["unit"] == "iters/s"
    assert out["value"] > 0 and out["ms_per_step"] > 0 and out["higher_is_better"] is True
    assert out["value"] == pytest.approx(2 * 3 / (out["ms_per_step"] * 3e-3), rel=1e-6)      # whole-job camera-steps / s


@pytest.mark.gpu
def test_bench_exchange_path_through_rccl_in_a_group_of_one(cuda):
    """What a one-GPU box can drive of the N > 1 path through the REAL transport: QED_BENCH_RCCL_SELF=1 initialises the `nccl`
    (= RCCL) backend with one rank and sends every step through the split graphs and the compact exchange -- the AVG
    all-reduce of the geometry gradients, all_gather_into_tensor of the colour-gradient message, both asynchronous and
    waited for on the compute stream between graph replays, the barriers and the rank agreement.  With one rank the
    exchange is the identity, so the step must train exactly as the single-GPU step does."""
    env = dict(os.environ, QED_BENCH_RCCL_SELF="1", QED_BENCH_COLLECTIVE_TIMEOUT="120")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "QED_BENCH_REHEARSE"):
        env.pop(k, None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
            "--no-cpu-baseline", "--no-api-path", "--gaussians", "100000", "--width", "960", "--height", "540"]
    outs = []
    # default: the whole step, collectives included, as ONE captured graph (gather behind the backward pass); =1: the gather
    # ahead of the projection backward; =0: three graphs around eager collectives (the fallback when a capture fails)
    early = dict(env, QED_BENCH_DP_ONE_GRAPH="1")
    three = dict(env, QED_BENCH_DP_ONE_GRAPH="0")
    for e in (three, {k: v for k, v in env.items() if k != "QED_BENCH_RCCL_SELF"}, early, env):
        p = subprocess.run(base, env=e, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-4000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]
        outs.append((json.loads(lines[0]), p.stderr))
    (o_self, err_self), (o_single, _), (o_one, err_one), (o_def, err_def) = outs
    assert "ONE hipGraph" in o_one["config"]["dispatch"] and "ahead of the projection" in o_one["config"]["dispatch"], err_one[-2000:]
    assert "ONE hipGraph" in o_def["config"]["dispatch"] and "behind the backward" in o_def["config"]["dispatch"], err_def[-2000:]
    for o in (o_one, o_def):
        c = o["config"]["intersections_after_timed_steps"]
        assert abs(c - o_single["config"]["intersections_after_timed_steps"]) <= 2e-3 * c
    assert "three hipGraphs" in o_self["config"]["dispatch"], err_self[-2000:]      # the N > 1 dispatch, captured
    assert "RCCL" in o_self["config"]["parallelism"] and o_single["config"]["parallelism"] == "single"
    assert o_self["n_gpus"] == 1 and o_self["value"] > 0
    # the same training: the list length after the timed steps is a fingerprint of the parameters
    assert o_self["config"]["intersections"] == o_single["config"]["intersections"]
    a, b = o_self["config"]["intersections_after_timed_steps"], o_single["config"]["intersections_after_timed_steps"]
    assert abs(a - b) <= 2e-3 * b, (a, b)


@pytest.mark.gpu
def test_two_rank_exchanges_agree_and_an_overflow_on_one_rank_skips_the_step_on_all(cuda):
    """scripts/dp_rehearsal.py on two gloo ranks that share cuda:0: the exchange forms (chunked, compact, views into the optimiser, overlapped, early gather, sparse message) leave the parameters the plain
    all-reduce + step leaves, replicas stay bit-identical -- also through a step in which ONE rank's frame overflows its
    intersection buffer: that rank renders an empty frame and every rank skips the update (the overflow word travels with
    the exchange; parallel.py), then all train on."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, QED_BENCH_REHEARSE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "scripts", "dp_rehearsal.py")]
    p = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    assert p.stdout.count("step skipped on this rank: True") == 4, p.stdout[-3000:]       # two ranks x two exchange forms
    # the sparse colour-gradient message (rows of the Gaussians a rank saw) trains like the dense one, on both ranks
    assert p.stdout.count("sparse colour-gradient message") == 2 and "== plain: False" not in p.stdout, p.stdout[-3000:]


def test_launcher_enforces_its_wall_limit_and_reaps_its_children(tmp_path):
    """launch_ranks() must not wait for a wedged rank until the driver's own limit: the process group it started is
    terminated at the wall limit and the exit status is non-zero.  (CPU: the 'ranks' here are a stand-in that sleeps.)"""
    import signal
    import textwrap
    import time
    stub = tmp_path / "torch" / "distributed"
    stub.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (stub / "__init__.py").write_text("")
    pidfile = tmp_path / "child.pid"
    # stands in for `python -m torch.distributed.run ...`: a launcher that starts one child and never returns
    (stub / "run.py").write_text(textwrap.dedent(f"""
        import os, subprocess, sys, time
        c = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"])
        open({str(pidfile)!r}, "w").write(f"{{os.getpid()}} {{c.pid}}")
        time.sleep(600)
    """))
    driver = textwrap.dedent(f"""
        import sys, types
        sys.path.insert(0, {ROOT!r})
        sys.argv = ["bench.py", "--gpus", "2", "--launch-timeout", "3"]
        import importlib.util
        spec = importlib.util.spec_from_file_location("bench_under_test", {os.path.join(ROOT, "bench.py")!r})
        import os
        os.environ["PYTHONPATH"] = {str(tmp_path)!r}          # the children import the stand-in launcher, not torch's
        os.environ.pop("QED_BENCH_LAUNCH_TIMEOUT", None)
        b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
        sys.exit(b.launch_ranks(b.parse()))
    """)
    t0 = time.time()
    p = subprocess.run([sys.executable, "-c", driver], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 124, (p.returncode, p.stderr[-2000:])
    assert time.time() - t0 < 60
    assert "terminating process group" in p.stderr
    pids = [int(x) for x in pidfile.read_text().split()]
    time.sleep(0.5)
    for pid in pids:                                               # launcher and its child are gone
        try:
            state = open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()[0]
        except (FileNotFoundError, ProcessLookupError):
            continue
        if state != "Z":                                           # (an orphaned zombie only waits for init to reap it)
            os.kill(pid, signal.SIGKILL)
            raise AssertionError(f"process {pid} (state {state}) survived the launcher's wall limit")
