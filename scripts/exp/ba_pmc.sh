#!/bin/bash
# SQ counters of the BA kernels at ba4 size (tuning aid): one pass, 8 SQ counters
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/prof_ba_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU \
  --output-format csv -d "$OUT/pmc" -- python3 "$R/bench.py" --workload ${1:-ba4} --steps 2 --warmup 1 --no-cpu > "$OUT/bench.json" 2> "$OUT/err.log" || { tail -5 "$OUT/err.log"; exit 1; }
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
for f in glob.glob(os.path.join(out, "pmc", "**", "*_counter_collection.csv"), recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[(r["Kernel_Name"][:36], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        if not k.startswith("__amd"):
            print(f"{k:38s} {c:22s} n={len(v):3d} avg={sum(v)/len(v):14.1f}")
PY
