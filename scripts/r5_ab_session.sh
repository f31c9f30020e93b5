#!/bin/bash
# A/B session on the GPU box: the GPU test suite on the product library, then scripts/composite_bench.py on every variant
# library named on the command line ("default" = the product library).  scripts/r5_ab_session.sh OUTDIR [notests] v1 v2 ...
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; shift
mkdir -p $O
cd $R
if [ "${1:-}" = notests ]; then shift; else
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
  rc=$?
  tail -5 $O/tests.log
  echo "tests rc=$rc"
  if [ $rc -ge 124 ]; then exit $rc; fi
fi
for v in "$@"; do
  if [ $v = default ]; then unset QED_SPLAT_LIB; else export QED_SPLAT_LIB=$R/qed_splatter_amd/lib/libqed_splat_$v.so; fi
  echo "== $v" | tee -a $O/cb.txt
  timeout -k 10 300 python scripts/composite_bench.py 40 2>> $O/cb.err | tee -a $O/cb.txt
  rc=${PIPESTATUS[0]}
  if [ $rc -ge 124 ]; then echo "bench $v killed rc=$rc"; exit $rc; fi
done
