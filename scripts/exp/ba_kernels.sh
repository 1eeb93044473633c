#!/bin/bash
# per-kernel averages of one BA workload (tuning aid):  gpurun -- bash scripts/exp/ba_kernels.sh ba4
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
W=${1:-ba4}
OUT=$R/gpurun_out/ba_kernels_$W
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$R/bench.py" --workload $W --no-cpu --steps 5 --warmup 1 | grep -o "\"ms_per_step\": [0-9.]*"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
for r in csv.DictReader(open(f)):
    print(r["Name"][:40], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
