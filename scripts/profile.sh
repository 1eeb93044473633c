#!/bin/bash
# Collects the rocprofv3 evidence behind bench.py's numbers.  Run on the GPU box:
#   gpurun -- bash scripts/profile.sh <tag>
# Results land in gpurun_out/prof_<tag>/ ; copy the summaries into profiles/.
set -e -o pipefail
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 1. per-kernel time of the default bench (config 2) -- one pass, trace + stats only
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_config2" -- \
  python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu --no-large --no-pcie > "$OUT/bench_config2.json" 2> "$OUT/kt_config2.err"
# 2. the same for the robust workload (k_pass launches)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_config5" -- \
  python3 "$R/bench.py" --workload config5 --steps 20 --warmup 3 --no-cpu --no-large --no-pcie > "$OUT/bench_config5.json" 2> "$OUT/kt_config5.err"
# 3. batched windows
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_config3" -- \
  python3 "$R/bench.py" --workload config3 --steps 5 --warmup 1 --no-cpu > "$OUT/bench_config3.json" 2> "$OUT/kt_config3.err"
# 3b. the config-4-size launches behind `roofline_large` (one workload per trace, so that the
#     per-kernel averages are not a mix of sizes)
for WL in config4 config4r; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_$WL" -- \
    python3 "$R/bench.py" --workload $WL --steps 8 --warmup 2 --no-cpu --no-pcie > "$OUT/bench_$WL.json" 2> "$OUT/kt_$WL.err"
done
# 4. HBM traffic counters at config-4 size, one counter per pass (FETCH_SIZE and WRITE_SIZE do not fit together)
for C in FETCH_SIZE WRITE_SIZE; do
  for WL in config2 config5 config3; do
    timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_${C}_${WL}" -- \
      python3 "$R/bench.py" --workload $WL --steps 3 --warmup 1 --no-cpu --no-large --no-pcie > "$OUT/bench_pmc_${C}_${WL}.json" 2> "$OUT/pmc_${C}_${WL}.err"
  done
  timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_${C}_config4" -- \
    python3 "$R/bench.py" --workload config4 --steps 3 --warmup 1 --no-cpu --no-pcie > "$OUT/bench_pmc_${C}.json" 2> "$OUT/pmc_${C}.err"
  timeout -s KILL 200 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_${C}_config4r" -- \
    python3 "$R/bench.py" --workload config4r --steps 3 --warmup 1 --no-cpu --no-pcie > "$OUT/bench_pmc_${C}_r.json" 2> "$OUT/pmc_${C}_r.err"
done
python3 "$R/scripts/summarize_profile.py" "$OUT" "$OUT/traffic.json" > "$OUT/SUMMARY.md"
cat "$OUT/SUMMARY.md"
