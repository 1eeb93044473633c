#!/bin/bash
# rocprofv3 evidence for the device-resident track gather (bench.py --workload tracks2 | tracks4).
# Run on the GPU box:   gpurun -- bash scripts/profile_tracks.sh <tag>
set -e -o pipefail
TAG=${1:-tracks}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for W in tracks2 tracks4; do
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$W" -- \
    python3 "$R/bench.py" --workload $W --steps 20 --warmup 3 --no-cpu > "$OUT/bench_$W.json" 2> "$OUT/kt_$W.err"
done
# HBM traffic of the gather kernels at config-4 size, one counter per pass
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_${C}_tracks4" -- \
    python3 "$R/bench.py" --workload tracks4 --steps 3 --warmup 1 --no-cpu > "$OUT/bench_pmc_${C}.json" 2> "$OUT/pmc_${C}.err"
done
python3 "$R/scripts/summarize_profile.py" "$OUT" "$OUT/traffic.json" > "$OUT/SUMMARY.md"
cat "$OUT/SUMMARY.md"
