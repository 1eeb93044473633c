#!/bin/bash
# keep the adapter outputs of a standalone run and of a run after test_gpu_tracks, and compare per mode
rm -rf gpurun_out/btA gpurun_out/btB
timeout -k 10 300 python -u -m pytest tests/test_host_adapter.py -m gpu -q --timeout 120 --timeout-method thread --tb=no --basetemp=gpurun_out/btA 2>&1 | tail -1
timeout -k 10 300 python -u -m pytest tests/test_gpu_tracks.py tests/test_host_adapter.py -m gpu -q --timeout 120 --timeout-method thread --tb=no --basetemp=gpurun_out/btB 2>&1 | tail -1
python3 - <<'PY'
import glob, os
for a in sorted(glob.glob("gpurun_out/btA/test_track_mirror*/*.bin")):
    b = a.replace("btA", "btB")
    if not os.path.exists(b):
        print("missing", b); continue
    da, db = open(a, "rb").read(), open(b, "rb").read()
    if da != db:
        first = next(i for i in range(min(len(da), len(db))) if da[i] != db[i])
        print("DIFF", a, "len", len(da), len(db), "first differing byte", first, "of which differing bytes", sum(x != y for x, y in zip(da, db)))
    else:
        print("same", a)
PY
