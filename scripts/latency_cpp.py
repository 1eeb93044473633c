"""Latency of the reference-side call from C++ (host/adapter_check latency): the adapter's refine()
(the reference's own chain walk on the host + mvo_hip_refine) and mvo_hip_refine alone, at the
reference's shipped defaults and at BASELINE configs[0]."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))
from mvo_amd import scene

EXE = os.path.join(ROOT, "monocular-visual-odometry_amd", "host", "adapter_check")
with tempfile.TemporaryDirectory() as d:
    for label, w, kw in (("defaults N=3 M=2 60 feat", scene.make_window(3, 2, 60, 7, span="full", keep_tracks=True), {}),
                         ("defaults + Cauchy", scene.make_window(3, 2, 60, 7, span="full", keep_tracks=True), dict(use_loss_function=True)),
                         ("N=10 M=5 300 feat", scene.make_window(10, 5, 300, 3, span="full", keep_tracks=True), {}),
                         ("N=30 M=5 5000 landmarks (config 2)", scene.make_window(30, 5, 5000, 1002, span=("sliding", 20), keep_tracks=True), {})):
        tracks, out = os.path.join(d, "t.bin"), os.path.join(d, "o.txt")
        # the reference's shipped solver options (mvo/cfg/mvo.cfg): 50 iterations, 1e-6 / 1e-10 / 1e-8
        scene.write_track_file(w, tracks, max_num_iterations=50, function_tolerance=1e-6, gradient_tolerance=1e-10,
                               parameter_tolerance=1e-8, **kw)
        subprocess.check_call([EXE, "latency", tracks, out])
        print(label + ": " + open(out).read().strip(), flush=True)
