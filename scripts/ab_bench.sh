#!/bin/bash
# development: the same workloads with two builds of the library (MVO_HIP_LIB_PATH)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for LIB in "$@"; do
  for WL in config3 config2; do
    ST=20; [ $WL = config3 ] && ST=5
    MVO_HIP_LIB_PATH=$R/monocular-visual-odometry_amd/$LIB timeout -k 10 300 python bench.py --workload $WL --steps $ST --warmup 3 --no-cpu --no-large > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$LIB $WL failed"; tail -3 gpurun_out/ab.err; exit 1; }
    python - <<PY
import json
b=json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
print("$LIB", "$WL", "ms/step", b["ms_per_step"], "device_ms", b["device_ms_per_step"])
PY
  done
done
