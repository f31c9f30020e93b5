#!/bin/bash
# One profiling session of the round-5 tree on the GPU box (run through gpurun): every rocprofv3 run is its own process with
# the program right behind `--`; counters in passes of their own (no trace domains beside --pmc).
#   scripts/r5_profile_session.sh [kt|k7|rccl|de|all]
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5p
mkdir -p $O
WHAT=${1:-all}
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-api-path"
PB="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-api-path"
if [ $WHAT = kt ] || [ $WHAT = all ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $BENCH > $O/kt.json 2> $O/kt.err
  echo "kernel trace done" ; tail -1 $O/kt.err
  python3 $R/scripts/kstats.py $O/kt/kt_kernel_stats.csv 0.05 > $O/kt_stats.txt
fi
if [ $WHAT = k7 ] || [ $WHAT = all ]; then
  # where K7's SIMD cycles go: issue / wait split, transcendental and LDS shares (three passes of 8 SQ counters at most)
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE \
      -d $O/pmc1 -o pmc1 -- $PB > $O/pmc1.json 2> $O/pmc1.err ; echo "pmc1 done"
  rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_LEVEL_LDS \
      -d $O/pmc2 -o pmc2 -- $PB > $O/pmc2.json 2> $O/pmc2.err ; echo "pmc2 done"
  rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM \
      -d $O/pmc3 -o pmc3 -- $PB > $O/pmc3.json 2> $O/pmc3.err ; echo "pmc3 done"
  for p in pmc1 pmc2 pmc3; do python3 $R/scripts/pmc_db_summary.py $(ls $O/$p/*.db | head -1) composite > $O/$p.txt 2>&1; done
  rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o fetch -- $PB > $O/fetch.json 2> $O/fetch.err ; echo "fetch done"
  rocprofv3 --pmc WRITE_SIZE -d $O/write -o write -- $PB > $O/write.json 2> $O/write.err ; echo "write done"
  python3 $R/scripts/pmc_to_json.py traffic $(ls $O/fetch/*.db | head -1) $(ls $O/write/*.db | head -1) $O/hbm_traffic_pmc.json "round-5 tree, bench.py --steps 10 --warmup 3"
  python3 $R/scripts/pmc_to_json.py valu $(ls $O/pmc1/*.db | head -1) $O/valu_issue_pmc.json "round-5 tree, bench.py --steps 10 --warmup 3"
fi
if [ $WHAT = rccl ] || [ $WHAT = all ]; then
  # the N > 1 dispatch (three graphs + eager collectives through RCCL) in a group of ONE rank, at config B
  QED_BENCH_RCCL_SELF=1 python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-api-path > $O/bench_rccl_self.json 2> $O/bench_rccl_self.err
  python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-api-path > $O/bench_single.json 2> $O/bench_single.err
  python3 -c "
import json
a=json.load(open('$O/bench_rccl_self.json')); b=json.load(open('$O/bench_single.json'))
print('rccl-self', a['ms_per_step'], a['config']['dispatch']); print('single   ', b['ms_per_step'], b['config']['dispatch'])"
fi
if [ $WHAT = de ] || [ $WHAT = all ]; then
  ST="python3 $R/scripts/stage_times.py --fused-sh --async-m --iters 6"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktD -o ktD -- $ST --gaussians 5000000 --width 1920 --height 1080 --seed 7 > $O/stage_D.txt 2> $O/stage_D.err
  python3 $R/scripts/kstats.py $O/ktD/ktD_kernel_stats.csv 0.05 > $O/ktD_stats.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktE -o ktE -- $ST --gaussians 2000000 --width 4096 --height 2160 --seed 9 > $O/stage_E.txt 2> $O/stage_E.err
  python3 $R/scripts/kstats.py $O/ktE/ktE_kernel_stats.csv 0.05 > $O/ktE_stats.txt
  rocprofv3 --pmc FETCH_SIZE -d $O/fetchD -o fetchD -- $ST --gaussians 5000000 --width 1920 --height 1080 --seed 7 > /dev/null 2> $O/fetchD.err
  rocprofv3 --pmc WRITE_SIZE -d $O/writeD -o writeD -- $ST --gaussians 5000000 --width 1920 --height 1080 --seed 7 > /dev/null 2> $O/writeD.err
  python3 $R/scripts/pmc_db_summary.py $(ls $O/fetchD/*.db | head -1) qed > $O/fetchD.txt 2>&1
  python3 $R/scripts/pmc_db_summary.py $(ls $O/writeD/*.db | head -1) qed > $O/writeD.txt 2>&1
fi
# the big databases stay on the box: only the summaries travel back
rm -rf $O/*/*.db $O/kt/*.csv.bak 2>/dev/null
find $O -name "*_kernel_trace.csv" -size +20M -delete
ls $O
