#!/bin/bash
# quick A/B numbers while tuning k_lm: ms per solve of the main workloads (no CPU leg, no large leg)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
for WL in config2 config5 config3 config4 config4r; do
  ST=20; [ $WL = config3 ] && ST=5; [ $WL = config4 ] && ST=8; [ $WL = config4r ] && ST=8
  timeout -k 10 300 python bench.py --workload $WL --steps $ST --warmup 3 --no-cpu --no-large > gpurun_out/qb_$WL.json 2> gpurun_out/qb_$WL.err || { echo "$WL failed"; tail -5 gpurun_out/qb_$WL.err; exit 1; }
  python - <<PY
import json
b=json.loads(open("gpurun_out/qb_$WL.json").read().strip().splitlines()[-1])
print("$WL", "ms/step", b["ms_per_step"], "value", b["value"], "device_ms", b["device_ms_per_step"], "pcie_refine_ms", b.get("refine_ms_pcie_inclusive"), "pageable", b.get("refine_ms_pcie_inclusive_pageable"))
PY
done
python scripts/latency_defaults.py 2>/dev/null | tail -4
