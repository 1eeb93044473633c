#!/bin/bash
# chunk-count sweep of the fold / LM overlap for batched windows (tuning aid)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for C in 2 3 4 5 6 8; do
  echo "== chunks=$C"
  MVO_HIP_OVERLAP_CHUNKS=$C timeout -k 10 120 python3 "$R/bench.py" --workload config3 --steps 20 --warmup 3 --no-cpu 2>/dev/null | python3 -c "import json,sys; b=json.loads(sys.stdin.read()); print(b['ms_per_step'], b['device_ms_per_step'], b['value'])" || exit 1
done
