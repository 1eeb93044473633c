#!/bin/bash
# Kernel-trace of the default bench; prints every k_fold / k_lm duration in launch order.
set -e
export TMPDIR=/tmp
OUT=gpurun_out/fold_trace
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-large > $OUT/bench.json 2> $OUT/bench.err
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/fold_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp'])
for r in rows:
    n = r['Kernel_Name'][:24]
    print(f"{(int(r['Start_Timestamp'])-t0)/1e3:12.1f} us  {n:26s} {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.2f} us  grid={r.get('Grid_Size_X', r.get('Grid_Size'))} wg={r.get('Workgroup_Size_X', r.get('Workgroup_Size'))}")
PY
