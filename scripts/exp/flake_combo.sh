#!/bin/bash
# which preceding test module makes test_host_adapter's TrackMirror case differ?
for pre in test_gpu_ba.py test_gpu_ba_partitioned.py test_gpu_epipolar.py test_gpu_fullsize.py test_gpu_parity.py test_gpu_partitioned.py test_gpu_scaler.py test_gpu_tracks.py; do
  r=$(timeout -k 10 300 python -u -m pytest tests/$pre tests/test_host_adapter.py -m gpu -q --timeout 120 --timeout-method thread --tb=no 2>&1 | tail -1)
  echo "$pre: $r"
done
