#!/bin/bash
# fold tiling granularity (MVO_HIP_WAVES_PER_CU) on the large workloads, kernel averages from the trace:
#   gpurun -- bash scripts/exp/fold_waves_sweep.sh <workload> <n> [<n> ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
WL=$1; shift
for N in "$@"; do
  OUT=$R/gpurun_out/wpc_${WL}_$N; mkdir -p "$OUT"
  cd /tmp && export TMPDIR=/tmp
  export MVO_HIP_WAVES_PER_CU=$N
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- \
    python3 "$R/bench.py" --workload $WL --steps 10 --warmup 3 --no-cpu --no-large --no-pcie > "$OUT/bench.json" 2> "$OUT/err.log" || { tail -5 "$OUT/err.log"; exit 1; }
  F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
  echo "== waves per CU $N"
  python3 - "$F" "$OUT/bench.json" <<'PY'
import csv, sys, json
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["AverageNs"]) > 5000:
        print(f'  {r["Name"][:50]:50s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:8.2f}')
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("  ms/step", d["ms_per_step"])
PY
done
