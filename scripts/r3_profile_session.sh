#!/bin/bash
# One profiling session of the round-3 tree on the GPU box (run through gpurun): every rocprofv3 run is its own process with
# the program right behind `--`; counters in passes of their own (no trace domains beside --pmc).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-api-path"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $BENCH > $O/kt.json 2> $O/kt.err
echo "kernel trace done" ; tail -1 $O/kt.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/api -o api -- python3 $R/scripts/api_path_bench.py 50 qed > $O/api.log 2>&1
echo "api trace done"
PB="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-api-path"
rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o fetch -- $PB > $O/fetch.json 2> $O/fetch.err ; echo "fetch done"
rocprofv3 --pmc WRITE_SIZE -d $O/write -o write -- $PB > $O/write.json 2> $O/write.err ; echo "write done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/valu -o valu -- $PB > $O/valu.json 2> $O/valu.err ; echo "valu done"
cd $R
python scripts/kstats.py $O/kt/kt_kernel_stats.csv 0.05 > $O/kt_stats.txt
python scripts/kstats.py $O/api/api_kernel_stats.csv 0.05 > $O/api_stats.txt
QED_SPLAT_LIB=qed_splatter_amd/lib/libqed_splat_stats.so python scripts/composite_stats.py > $O/composite_stats.txt 2>&1
ls $O $O/fetch $O/write $O/valu
