#!/bin/bash
# experiment: effect of the tile size on the fold / pass kernels at config-4 size
for T in 128 256 512 1024 2048 4096; do
  for WL in config4 config4r; do
    MVO_HIP_TILE=$T timeout -k 10 200 python bench.py --workload $WL --no-cpu --steps 5 --warmup 1 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read())
print('tile', $T, '$WL', 'ms/step', b['ms_per_step'], 'fold_us', b['fold_kernel_us'], 'roofline', b['roofline']['achieved'], b['roofline']['avg_launch_us'])"
  done
done
