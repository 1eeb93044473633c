"""Per-step HIP-event durations of the config-2 fold launch (is the average an outlier's doing?)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from mvo_amd import capi
ws, opt, desc = bench.build_workload("config2", 0)
s = capi.Solver(0)
s.upload(ws, opt)
for i in range(3):
    s.solve(); s.download()
vals = []
for i in range(24):
    timed = i % 4 == 0
    s.set_profiling(timed)
    s.solve(); s.download(copy=False)
    if timed:
        tm = s.timing()
        vals.append((round(tm["assembly_first_seconds"] * 1e6, 2), round(tm["device_seconds"] * 1e6, 1)))
print("every 4th:", vals)
s.set_profiling(True)
vals = []
for i in range(8):
    s.solve(); s.download(copy=False)
    tm = s.timing()
    vals.append((round(tm["assembly_first_seconds"] * 1e6, 2), round(tm["device_seconds"] * 1e6, 1)))
print("every step:", vals)
