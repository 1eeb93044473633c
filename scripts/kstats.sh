#!/bin/bash
# per-kernel averages of one workload: gpurun -- bash scripts/kstats.sh <workload> [steps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WL=${1:-config5}; ST=${2:-20}
OUT=$R/gpurun_out/ks_$WL
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- \
  python3 "$R/bench.py" --workload $WL --steps $ST --warmup 3 --no-cpu --no-large --no-pcie > "$OUT/bench.json" 2> "$OUT/err.log" || { tail -5 "$OUT/err.log"; exit 1; }
F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:8.2f}  max {float(r["MaxNs"])/1e3:8.2f}')
PY
