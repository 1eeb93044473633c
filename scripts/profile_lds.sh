#!/bin/bash
# LDS and issue counters of the device-resident LM kernel (SURVEY.md 8d "Counters").  One pass,
# 7 SQ counters (the block holds 8).  Run on the GPU box:  gpurun -- bash scripts/profile_lds.sh <tag>
set -e -o pipefail
TAG=${1:-lds}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for WL in config2 config3; do
  timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS \
    --output-format csv -d "$OUT/pmc_sq_$WL" -- \
    python3 "$R/bench.py" --workload $WL --steps 3 --warmup 1 --no-cpu --no-large > "$OUT/bench_sq_$WL.json" 2> "$OUT/pmc_sq_$WL.err"
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
print("| workload | kernel | dispatches | counter | avg per dispatch |\n|---|---|---|---|---|")
for d in sorted(glob.glob(os.path.join(out, "pmc_sq_*"))):
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:48], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            if not k.startswith("__amd"):
                print(f"| {os.path.basename(d)[7:]} | `{k}` | {len(v)} | {c} | {sum(v)/len(v):.1f} |")
PY
