"""Condenses a scripts/profile.sh output directory into one markdown summary."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "monocular-visual-odometry_amd"))
from mvo_amd.srchash import source_hash  # noqa: E402

out = sys.argv[1]
print(f"# rocprofv3 summary ({os.path.basename(out)})\n")
print("Each trace is ONE workload, the bench's timed region and its warm-up only (`--no-large --no-pcie` on config 2 / 5:\n"
      "the host-buffer legs launch the same fold kernel cold behind a copy, 12-14 us, or reading pinned host memory\n"
      "in place, 111 us, and would be averaged into the kernel's line; config 3 / 4 / 4r traced on their own).\n"
      "`avg_launch_us` in the bench lines below was measured with HIP events UNDER the profiler, which adds ~10 us\n"
      "to every event interval; outside the profiler the config-2 interval reads 9.8 us (events cost ~2.4 us around\n"
      "a 7.4-us kernel) and the config-4-size launches agree with the trace within 1 %: `profiles/*/bench_*_unprofiled.json`.\n")
for d in sorted(glob.glob(os.path.join(out, "kt_*"))):
    if not os.path.isdir(d):
        continue
    name = os.path.basename(d)[3:]
    print(f"## kernel-trace --stats: bench.py --workload {name}\n")
    bj = os.path.join(out, f"bench_{name}.json")
    if os.path.exists(bj) and os.path.getsize(bj):
        b = json.loads(open(bj).read().strip().splitlines()[-1])
        print(f"bench line: value={b['value']} {b['unit']}, ms_per_step={b['ms_per_step']}, roofline={json.dumps(b.get('roofline'))}\n")
    for f in glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True):
        print("| kernel | calls | total ns | avg ns | % | min ns | max ns |\n|---|---|---|---|---|---|---|")
        for r in csv.DictReader(open(f)):
            print(f"| `{r['Name'][:70]}` | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {r['Percentage']} | {r['MinNs']} | {r['MaxNs']} |")
        print()
    for f in glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True):
        acc = collections.defaultdict(list)
        meta = {}
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:70]
            acc[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            meta[k] = (r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Grid_Size"), r.get("Workgroup_Size"))
        print("| kernel | median ns | vgpr | agpr | sgpr | lds | grid | wg |\n|---|---|---|---|---|---|---|---|")
        for k, v in acc.items():
            v.sort()
            m = meta[k]
            print(f"| `{k}` | {v[len(v)//2]} | {m[0]} | {m[1]} | {m[2]} | {m[3]} | {m[4]} | {m[5]} |")
        print()
print("## PMC passes (one counter per run), per dispatch averages\n")
traffic = collections.defaultdict(dict)  # workload -> kernel -> {fetch_bytes, write_bytes}
print("FETCH_SIZE / WRITE_SIZE are in KiB. On gfx950 FETCH_SIZE counts 64 B per 128-B request for wide\ncoalesced reads (MI355X_MICROARCH.md, HBM section): corrected read bytes = 2 x FETCH_SIZE x 1024.\n")
print("| pass | kernel | dispatches | counter | avg value |\n|---|---|---|---|---|")
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in acc.items():
            if k.startswith("__amd"):
                continue
            print(f"| {os.path.basename(d)} | `{k}` | {len(v)} | {c} | {sum(v)/len(v):.1f} |")
            wl = os.path.basename(d).split("_")[-1]
            kern = k.replace("void ", "").split("(")[0]
            avg_kib = sum(v) / len(v)
            if c == "FETCH_SIZE":  # gfx950: 64 B counted per 128-B request of a wide coalesced read
                traffic[wl].setdefault(kern, {})["fetch_bytes"] = int(2 * avg_kib * 1024)
            elif c == "WRITE_SIZE":
                traffic[wl].setdefault(kern, {})["write_bytes"] = int(avg_kib * 1024)
for wl in traffic:  # bench.py prints `traffic: null` once a kernel's source no longer matches
    for kern in traffic[wl]:
        traffic[wl][kern]["source_hash"] = source_hash(kern)
if len(sys.argv) > 2:
    json.dump({"note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate runs); "
                       "FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction",
               "per_workload": traffic}, open(sys.argv[2], "w"), indent=1, sort_keys=True)
