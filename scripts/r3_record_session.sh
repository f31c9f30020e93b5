#!/bin/bash
# The records of the round-3 tree that are not profiles: full GPU test tier, the default bench line, both parity sweeps, the
# per-stage times of configs D and E (run through gpurun; outputs under gpurun_out/r3r/).
set -u
O=gpurun_out/r3r
mkdir -p $O
python -m pytest tests -q -m gpu > $O/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -2 $O/gputests.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python scripts/parity_sweep.py 100 240 > $O/sweep_fused.txt 2>&1; echo "sweep fused rc=$?"; tail -1 $O/sweep_fused.txt
python scripts/parity_sweep_api.py 100 240 > $O/sweep_api.txt 2>&1; echo "sweep api rc=$?"; tail -1 $O/sweep_api.txt
python scripts/stage_times.py --gaussians 5000000 --width 1920 --height 1080 --iters 5 --fused-sh --async-m --seed 7 > $O/stage_D.txt 2>&1; tail -3 $O/stage_D.txt
python scripts/stage_times.py --gaussians 2000000 --width 4096 --height 2160 --iters 5 --fused-sh --async-m --seed 9 > $O/stage_E.txt 2>&1; tail -3 $O/stage_E.txt
