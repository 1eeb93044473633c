#!/bin/bash
# development loop on the GPU box: stamps, the GPU suite, config-2 kernel stats and the quick bench
# usage: gpurun -- bash scripts/dev_gpu.sh <tag> [notest]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; T=${1:-x}
timeout -k 10 200 python scripts/lm_stamps.py > gpurun_out/stamps_$T.txt 2>&1 || { tail -5 gpurun_out/stamps_$T.txt; exit 1; }
grep -A1 "^config2\|^defaults\|^config4 " gpurun_out/stamps_$T.txt
if [ "$2" != notest ]; then
  timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputest_$T.txt 2>&1
  rc=$?; tail -4 gpurun_out/gputest_$T.txt; [ $rc = 0 ] || exit $rc
fi
timeout -k 10 120 bash scripts/kstats.sh config2 20 > gpurun_out/ks_${T}_config2.txt 2>&1 || exit 1
cat gpurun_out/ks_${T}_config2.txt
timeout -k 10 500 bash scripts/quick_bench.sh > gpurun_out/qb_$T.txt 2>&1 || { tail gpurun_out/qb_$T.txt; exit 1; }
cat gpurun_out/qb_$T.txt
