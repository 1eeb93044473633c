#!/bin/bash
# rocprofv3 evidence for bench.py --workload scaler.  Run on the GPU box:
#   gpurun -- bash scripts/profile_scaler.sh <tag>
set -e -o pipefail
TAG=${1:-scaler}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_scaler" -- \
  python3 "$R/bench.py" --workload scaler --steps 20 --warmup 3 --no-cpu > "$OUT/bench_scaler.json" 2> "$OUT/kt_scaler.err"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_${C}_scaler" -- \
    python3 "$R/bench.py" --workload scaler --steps 3 --warmup 1 --no-cpu > "$OUT/bench_pmc_${C}.json" 2> "$OUT/pmc_${C}.err"
done
python3 "$R/scripts/summarize_profile.py" "$OUT" "$OUT/traffic.json" > "$OUT/SUMMARY.md"
cat "$OUT/SUMMARY.md"
