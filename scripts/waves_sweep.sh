#!/bin/bash
# experiment: wavefronts per CU of the observation passes (chunk size) at config-4 / config-3 size
for WPC in 8 12 16 24 32; do
  for WL in config4 config4r config3; do
    MVO_HIP_WAVES_PER_CU=$WPC timeout -k 10 200 python bench.py --workload $WL --no-cpu --steps 5 --warmup 1 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read())
print('waves/CU', $WPC, '$WL', 'ms/step', b['ms_per_step'], 'fold_us', b['fold_kernel_us'], 'roofline GB/s', b['roofline']['achieved'], 'us', b['roofline']['avg_launch_us'])"
  done
done
