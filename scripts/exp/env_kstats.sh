#!/bin/bash
# kernel averages of one workload under environment settings, same box:
#   gpurun -- bash scripts/exp/env_kstats.sh <workload> <steps> "" "A=1" "A=1 B=2" ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
WL=$1; ST=$2; shift 2
I=0
for E in "$@"; do
  I=$((I+1))
  OUT=$R/gpurun_out/env_${WL}_$I; mkdir -p "$OUT"
  cd /tmp && export TMPDIR=/tmp
  (
    for kv in $E; do export "$kv"; done
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- \
      python3 "$R/bench.py" --workload $WL --steps $ST --warmup 3 --no-cpu --no-large --no-pcie > "$OUT/bench.json" 2> "$OUT/err.log"
  ) || { tail -5 "$OUT/err.log"; exit 1; }
  F=$(find "$OUT" -name "*kernel_stats.csv" | head -1)
  echo "== [$E]"
  python3 - "$F" "$OUT/bench.json" <<'PY'
import csv, sys, json
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["AverageNs"]) > 5000:
        print(f'  {r["Name"][:50]:50s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  min {float(r["MinNs"])/1e3:8.2f}')
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("  ms/step", d["ms_per_step"])
PY
done
