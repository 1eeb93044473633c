"""Static instruction counts of k_lm between the stamps of the -DMVO_LM_STAMPS build (runs on the CPU:
hipcc cross-compiles), plus register / spill figures of the shipped build.  A single wavefront issues
one instruction per ~4.4 cycles whatever the dependences, so the count is the time."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "monocular-visual-odometry_amd")
names = ["prologue+pairs", "classes", "suffix", "products", "(b2)", "?", "decide", "finalize", "setup", "eliminate", "mcc", "plus", "publish", "tail", "tail2"]
subprocess.check_call(f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMVO_LM_STAMPS -S --cuda-device-only -o /tmp/mvo_stamps.s csrc/mvo_hip.hip 2>/dev/null", shell=True, cwd=PKG)
text = open("/tmp/mvo_stamps.s").read()
for nt in ("64", "256"):
    m = re.search(r"^_Z4k_lmILi" + nt + r".*?s_endpgm", text, re.S | re.M)
    seg, cur = [], []
    for l in m.group(0).splitlines():
        t = l.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_memtime":
            seg.append(cur); cur = []
        else:
            cur.append(op)
    seg.append(cur)
    print(f"k_lm<{nt}> (stamps build): instructions between stamps")
    for i, s in enumerate(seg):
        c = collections.Counter(s)
        print(f"  {i:2d} {len(s):5d}  agpr-moves {c['v_accvgpr_read_b32'] + c['v_accvgpr_write_b32']:4d}  sgpr-spill-moves {c['v_readlane_b32'] + c['v_writelane_b32']:4d}"
              f"  lds {sum(v for k, v in c.items() if k.startswith('ds_')):4d}  f64 {sum(v for k, v in c.items() if 'f64' in k and not k.startswith('v_cmp')):4d}"
              f"  cndmask {c['v_cndmask_b32_e64'] + c['v_cndmask_b32_e32']:4d}  mov {c['v_mov_b32_e32'] + c['v_mov_b64_e32']:4d}")
r = subprocess.run(f"/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c --cuda-device-only -Rpass-analysis=kernel-resource-usage -o /tmp/x.o csrc/mvo_hip.hip 2>&1 | grep -A12 'Function Name: .*k_lm' | grep -E 'VGPRs:|AGPRs|Spill|ScratchSize' | sed 's/\\[-Rpass.*//; s/.*remark: *//' | paste -sd' '", shell=True, cwd=PKG, capture_output=True, text=True)
print("shipped build:", r.stdout.strip())
