#!/bin/bash
# rocprofv3 kernel trace of the bundle-adjustment workloads.  Run on the GPU box:
#   gpurun -- bash scripts/profile_ba.sh <tag>
set -e -o pipefail
TAG=${1:-ba}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for W in ba2 ba5 ba4; do
  timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$W" -- \
    python3 "$R/bench.py" --workload $W --steps 5 --warmup 1 --no-cpu > "$OUT/bench_$W.json" 2> "$OUT/kt_$W.err"
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
for w in ("ba2", "ba5", "ba4"):
    f = max(glob.glob(os.path.join(out, f"kt_{w}", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    b = json.load(open(os.path.join(out, f"bench_{w}.json")))
    print(f"## {w}: ms_per_step={b['ms_per_step']} value={b['value']} roofline={json.dumps(b['roofline'])}")
    for r in csv.DictReader(open(f)):
        print(f"| `{r['Name'][:40]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} us | {r['Percentage']} % |")
PY
