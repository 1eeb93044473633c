#!/bin/bash
# A/B of library builds on one workload: gpurun -- bash scripts/exp/ab_kstats.sh <workload> <steps> <lib.so> [<lib.so> ...]
# (kstats.sh per build, each into its own directory)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
WL=$1; ST=$2; shift 2
for L in "$@"; do
  T=$(basename "$L" .so)
  OUT=$R/gpurun_out/ab_${WL}_$T
  mkdir -p "$OUT"
  cd /tmp && export TMPDIR=/tmp
  MVO_HIP_LIB_PATH="$R/$L" timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- \
    python3 "$R/bench.py" --workload $WL --steps $ST --warmup 3 --no-cpu --no-large --no-pcie > "$OUT/bench.json" 2> "$OUT/err.log" || { tail -5 "$OUT/err.log"; exit 1; }
  F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
  echo "== $T"
  python3 - "$F" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["AverageNs"]) > 5000:
        print(f'  {r["Name"][:50]:50s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:8.2f}')
PY
done
