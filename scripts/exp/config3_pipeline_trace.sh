#!/bin/bash
# kernel timeline of one config-3 batch under the staggered plan: gpurun -- bash scripts/exp/config3_pipeline_trace.sh [ENV=val ...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/pipe_trace; mkdir -p "$OUT"
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- \
  python3 "$R/bench.py" --workload config3 --steps 4 --warmup 2 --no-cpu --no-large --no-pcie > "$OUT/bench.json" 2> "$OUT/err.log" || { tail -5 "$OUT/err.log"; exit 1; }
F=$(find "$OUT" -name "*kernel_trace.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last batch: from the last fold launch that follows a gap
last = [i for i, r in enumerate(rows) if "k_fold" in r["Kernel_Name"]]
# find start of the final group of folds
i0 = last[-1]
while i0 > 0 and int(rows[i0]["Start_Timestamp"]) - int(rows[i0 - 1]["End_Timestamp"]) < 150000 and i0 - 1 >= 0 and ("k_fold" in rows[i0-1]["Kernel_Name"] or "k_lm" in rows[i0-1]["Kernel_Name"]):
    i0 -= 1
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    print(f'{(int(r["Start_Timestamp"])-t0)/1e3:9.1f} {(int(r["End_Timestamp"])-t0)/1e3:9.1f}  {(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3:8.1f} us  grid {r.get("Grid_Size_X","?"):>8s}  {r["Kernel_Name"][:60]}')
PY
