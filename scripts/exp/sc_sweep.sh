#!/bin/bash
# sweep of the scaler's workgroups-per-CU knobs (tuning aid)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for B in 2 4 6 8 16; do
  echo "== tracks per_cu=$B"
  MVO_SC_TRACKS_BLOCKS_PER_CU=$B timeout -k 10 120 python3 "$R/bench.py" --workload scaler --steps 20 --warmup 3 --no-cpu | python3 -c "import json,sys; b=json.loads(sys.stdin.read()); print(b['ms_per_step'], b['whole_run'], b['roofline']['avg_launch_us'])" || exit 1
done
