"""What ONE rank of an N-GPU run does, measured on one GPU: the pieces DESIGN.md section 6 builds its
predicted 1 -> 8 curve from.  NOT a multi-GPU measurement: no second GPU, no xGMI, the collective is a
world-of-one RCCL all-reduce (the library's own code path, nothing crossing a link).

  config3  (BASELINE configs[2], strong): rank 0's share of the 4096 windows, mvo_hip_solve + download
  config4  (configs[3], no loss):  rank 0's share of every pair's residual blocks, mvo_hip_solve_partitioned
  config4r (the same with 20 % outliers and the Cauchy loss)

gpurun -- python3 scripts/predict_scaling.py  ->  gpurun_out/predict_scaling.json"""
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))

from mvo_amd import capi, scene, sharding  # noqa: E402

FIXED10 = dict(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
WORLDS = (1, 2, 4, 8)


def median_ms(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return round(statistics.median(ts) * 1e3, 4)


out = {"note": "one GPU; per-rank share of an N-GPU run; world-of-one RCCL group", "rows": []}
with capi.Solver(0, profiling=False) as s:
    s.comm_init(capi.comm_unique_id(), 0, 1)
    for name, kw, loss in (("config4", {}, False), ("config4r", dict(outlier_fraction=0.2), True)):
        w = scene.config4(seed=1004, **kw)
        opt = capi.make_options(use_loss_function=loss, **FIXED10)
        for n in WORLDS:
            part = sharding.shard_observations(w, 0, n)
            s.upload([part], opt)

            def run():
                s.solve_partitioned_rccl()
                s.download(copy=False)
            ms = median_ms(run)
            blocks = int(part.pair_offsets[-1])
            out["rows"].append(dict(workload=name, world=n, residual_blocks_on_rank=blocks, ms_per_solve_rank0=ms))
            print(name, n, blocks, ms, flush=True)
        del w
    s.comm_destroy()

    total = 4096
    wins = [scene.config3_window(i) for i in range(total)]
    opt = capi.make_options(**FIXED10)
    for n in WORLDS:
        ids = sharding.shard_windows(total, 0, n)
        s.upload([wins[i] for i in ids], opt)

        def run3():
            s.solve()
            s.download(copy=False)
        ms = median_ms(run3, reps=20, warm=3)
        out["rows"].append(dict(workload="config3", world=n, windows_on_rank=len(ids), ms_per_batch_rank0=ms))
        print("config3", n, len(ids), ms, flush=True)

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "predict_scaling.json"), "w") as f:
    json.dump(out, f, indent=1)
