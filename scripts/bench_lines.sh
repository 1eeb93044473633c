#!/bin/bash
# The unprofiled bench lines, stamped phase tables and the C++ latency table that go with a profile:
#   gpurun -- bash scripts/bench_lines.sh <tag>     -> gpurun_out/lines_<tag>/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; T=${1:-x}; OUT=gpurun_out/lines_$T; mkdir -p $OUT
timeout -k 10 400 python bench.py > $OUT/bench_default.json 2> $OUT/default.err || { tail -3 $OUT/default.err; exit 1; }
for WL in config2 config3 config4 config4r config5; do
  timeout -k 10 300 python bench.py --workload $WL --steps 20 --warmup 3 --no-cpu --no-large --no-pcie > $OUT/bench_${WL}_unprofiled.json 2> $OUT/$WL.err || { tail -3 $OUT/$WL.err; exit 1; }
done
timeout -k 10 200 python scripts/lm_stamps.py > $OUT/lm_stamps_cycles.txt 2>&1 || exit 1
timeout -k 10 100 python scripts/lm4_stamps.py 4096 >> $OUT/lm_stamps_cycles.txt 2>&1 || exit 1
timeout -k 10 300 python scripts/latency_cpp.py > $OUT/latency_cpp.txt 2>&1 || { tail -3 $OUT/latency_cpp.txt; exit 1; }
python - $OUT <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), d["value"], d["unit"], "ms/step", d["ms_per_step"], "device", d.get("device_ms_per_step"), "roofline", d["roofline"]["frac"] if d.get("roofline") else None)
PY
cat $OUT/latency_cpp.txt
