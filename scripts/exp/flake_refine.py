"""Repeat the adapter's host-route refine while this (parent) process holds a live GPU context."""
import os, subprocess, sys, struct, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mvo_amd import scene, capi, tracks_capi
from _util import FIXED10
EXE = os.path.join(ROOT, "monocular-visual-odometry_amd", "host", "adapter_check")
w = scene.make_window(8, 3, 50, 8, span=("anchored", 2, 8), keep_tracks=True)
d = tempfile.mkdtemp()
tr = os.path.join(d, "t.bin")
scene.write_track_file(w, tr, use_loss_function=False, **FIXED10)
def once(tag):
    out = os.path.join(d, "o.bin")
    subprocess.check_call([EXE, "refine", tr, out])
    raw = open(out, "rb").read()
    got = np.frombuffer(raw, dtype="<f8", count=4 * w.N).reshape(w.N, 4)
    term, reason, iters = struct.unpack_from("<3i", raw, 32 * w.N)
    c0, c1 = struct.unpack_from("<2d", raw, 32 * w.N + 16)
    print(tag, "term", term, "reason", reason, "iters", iters, "cost %.6e -> %.6e" % (c0, c1), "vec0", got[0], flush=True)
for i in range(3):
    once("no parent context %d" % i)
solver = capi.Solver(0)
table = tracks_capi.TrackTable(solver, max_frames=8, max_features=512)
for i in range(12):
    if i % 3 == 0:  # some GPU work in the parent between the runs
        ws = [scene.config1()]
        solver.upload(ws, capi.make_options(**FIXED10)); solver.solve(); solver.download()
    once("parent context alive %d" % i)
