#!/bin/bash
# config 3 with the staggered fold / LM plan (MVO_HIP_PIPELINE) against fold-then-k_lm4, same box:
#   gpurun -- bash scripts/exp/config3_pipeline.sh      -> gpurun_out/pipe/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; OUT=gpurun_out/pipe; mkdir -p $OUT
run() {  # name, env...
  local n=$1; shift
  env "$@" timeout -k 10 200 python bench.py --workload config3 --steps 20 --warmup 3 --no-cpu --no-large --no-pcie > $OUT/$n.json 2> $OUT/$n.err || { tail -3 $OUT/$n.err; exit 1; }
  python - $OUT/$n.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "ms/step", d["ms_per_step"], "device", d.get("device_ms_per_step"))
PY
}
run plain A=1
run pipe MVO_HIP_PIPELINE=1
run pipe_mid256 MVO_HIP_PIPELINE=1 MVO_HIP_PIPELINE_MID=256
run pipe_mid512 MVO_HIP_PIPELINE=1 MVO_HIP_PIPELINE_MID=512
run plain2 A=1
