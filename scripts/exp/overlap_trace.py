"""Prints the kernel timeline of one config-3 solve from a rocprofv3 kernel trace (tuning aid)."""
import csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:12], r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))]
rows.sort()
t0 = rows[-20][0]
for s, e, k, q in rows[-20:]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} us  {k:14s} queue {q}")
