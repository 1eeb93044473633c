#!/usr/bin/env python3
"""bench.py -- throughput of libmvo_hip's sliding-window refinement on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload config2|config3|config4|config4r|config5|scaler|epipolar|tracks2|tracks4|ba2|ba4|ba5]

(ba*: the explicit-landmark bundle adjustment of include/mvo_hip_ba.h -- SURVEY.md 8f rank 1, no
reference counterpart -- on BASELINE.json's config shapes; see bench_ba below.)

A "step" is one complete refine() of the workload with the observations already resident in HBM
(mvo_hip_upload done before the timed region; mvo_hip_solve + mvo_hip_download inside it):
fold/moment pass or observation passes, the device-resident LM loop, results back on the host.

Default workload = BASELINE.json configs[1]: one synthetic 30-keyframe window, 5 000 landmarks,
~100k bearing observations, fp64, exactly 10 LM iterations (tolerances 0).  With N > 1 every rank
owns one such window (independent windows, no data-path collective): weak scaling.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts its N ranks ITSELF: the
parent spawns `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same flags>` as a
fresh child process before anything in the parent has touched the GPU (or imported torch), relays the
child's output and exits with its code.  Launched by torch.distributed.run directly (WORLD_SIZE set) it is
a rank.  The N > 1 default workload is BASELINE configs[2] (the same 4096 windows sharded, no collective);
`--workload config3|config4|config4r` runs that one workload at any N, N = 1 included, so that a 1/2/4/8
curve of one workload can be taken; the N = 1 default line (configs[1]) carries `scaling_base`, the
N > 1 default workload measured on this one GPU, and every N > 1 line carries `same_workload_one_gpu` and
`efficiency_vs_one_gpu`.

One JSON line on stdout (rank 0).  `roofline` describes the dominant observation-pass kernel of
the workload, timed with HIP events on the library's own stream inside the timed region;
`roofline_large` is the same kernel measured at config-4 size (~9.7M residual blocks, larger than
the 256 MiB Infinity Cache) because a 6 MB problem cannot exercise HBM; `cpu_baseline` is the CPU
oracle (a scalar port of the reference algorithm; the reference itself needs Ceres and cannot be
built here) timed on this node on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "monocular-visual-odometry_amd"))

BK_DEVICE = "cuda"  # where the bench's own bookkeeping tensors live ("cpu" in a gloo rehearsal)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md); ~6300 achievable


def plan(workload, world, weak=False):
    """Which BASELINE.json configuration a run measures and how it is spread over `world` GPUs.
    -> (workload, mode); mode: "single" (one GPU), "strong_windows" (the SAME 4096 config-3 windows
    block-partitioned over the ranks, no collective -- configs[2]), "partitioned" (ONE config-4 window,
    residual blocks range-partitioned, RCCL all-reduce inside the library -- configs[3]), "weak" (every
    rank its own copy of the workload; only on request).
    Default workload: configs[1] (config2) on one GPU -- the configuration the metric is quoted on --
    and configs[2] (config3), the configuration BASELINE.json names for 1/2/4/8 GPUs, beyond."""
    if workload is None:
        workload = "config2" if world == 1 else "config3"
    if world == 1:
        return workload, "single"
    if weak:
        return workload, "weak"
    if workload == "config3":
        return workload, "strong_windows"
    if workload in ("config4", "config4r"):
        return workload, "partitioned"
    return workload, "weak"  # config2 / config5 are single windows sized for one GPU


def build_workload(name, rank, world=1, mode="single"):
    from mvo_amd import capi, scene, sharding

    fixed10 = dict(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    seed_rank = rank if mode == "weak" else 0
    if name == "config2":
        ws = [scene.config2(seed=1002 + seed_rank)]
        opt = capi.make_options(**fixed10)
        desc = "config2: 1 window/GPU, N=30 M=5, 5000 landmarks, fixed 10 LM iterations, no loss"
    elif name == "config5":
        ws = [scene.config2(seed=1005 + seed_rank, outlier_fraction=0.2)]
        opt = capi.make_options(use_loss_function=True, **fixed10)
        desc = "config5: config2 data with 20% outliers, Cauchy(0.5) IRLS, fixed 10 LM iterations"
    elif name == "config3":
        n = int(os.environ.get("MVO_BENCH_WINDOWS", "4096"))
        if mode == "strong_windows":   # BASELINE configs[2]: the same n windows sharded over the ranks
            ids = sharding.shard_windows(n, rank, world)
            desc = f"config3: {n} independent windows sharded over {world} GPUs, N=10 M=5, 300 landmarks each, fixed 10 LM iterations"
        else:
            ids = range(seed_rank * n, seed_rank * n + n)
            desc = f"config3: {n} independent windows/GPU, N=10 M=5, 300 landmarks each, fixed 10 LM iterations"
        ws = [scene.config3_window(i) for i in ids]
        opt = capi.make_options(**fixed10)
    elif name == "config4":
        ws = [scene.config4(seed=1004 + seed_rank)]
        opt = capi.make_options(**fixed10)
        desc = "config4 (one GPU's worth): N=200 M=5, 985 pairs, ~9.7M residual blocks, fixed 10 LM iterations"
    elif name == "config4r":
        ws = [scene.config4(seed=1004 + seed_rank, outlier_fraction=0.2)]
        opt = capi.make_options(use_loss_function=True, **fixed10)
        desc = "config4 data with 20% outliers, Cauchy(0.5) IRLS, fixed 10 LM iterations"
    else:
        raise SystemExit(f"unknown workload {name}")
    return ws, opt, desc


def measured_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/traffic.json,
    written by scripts/profile.sh on the same workload).  None when no such pass was recorded -- or when
    the kernel's source has changed since (each entry carries the hash of the source files it was
    measured on): a stale figure is not reported."""
    from mvo_amd.srchash import source_hash

    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["per_workload"][workload]
    except (OSError, KeyError, ValueError):
        return None
    for name, rec in t.items():
        if name.startswith(kernel):
            if rec.get("source_hash") is None or rec.get("source_hash") != source_hash(name):
                return None
            return int(rec.get("fetch_bytes", 0) + rec.get("write_bytes", 0))
    return None


def pass_kernel(workload):
    # config 5: one window whose pass grid fits the chip in one round -- the passes run inside the persistent
    # robust solver (k_lm<256, 6>: ten passes, LM steps and grid barriers in one launch; its bytes per launch are
    # the ten passes' 24 B per residual block)
    if workload == "config5":
        return "k_lm<256, 6>"
    return "k_pass<true>" if workload == "config4r" else "k_fold<false, true, 0>"


def roofline_of(timing, workload=None):
    """achieved = algorithmic bytes per launch / average launch duration of the pass kernel."""
    n = max(1, timing["assembly_launches"])
    dur = timing["assembly_seconds"] / n
    if dur <= 0 or timing["assembly_bytes"] == 0:
        return None
    gbs = timing["assembly_bytes"] / dur / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": measured_traffic(workload, pass_kernel(workload)) if workload else None,
            "kernel": pass_kernel(workload) if workload else None,
            "bytes_per_launch": int(timing["assembly_bytes"]), "avg_launch_us": round(dur * 1e6, 3),
            "launches_timed": int(timing["assembly_launches"])}


def cpu_baseline(ws, opt_kw, use_loss):
    """The CPU oracle on the same workload (bounded: one window, the same fixed 10 iterations)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mvo_oracle_py as orc

    w = ws[0]
    opt = orc.make_options(use_loss_function=use_loss, **opt_kw)
    sample = w
    note = "the whole window"
    if w.n_residual_blocks > 600_000:  # keep the CPU leg to ~10-30 s: every 16th block of each pair
        import numpy as np
        from mvo_amd import scene

        offs, a0, a1 = [0], [], []
        for p in range(w.n_pairs):
            lo, hi = int(w.pair_offsets[p]), int(w.pair_offsets[p + 1])
            a0.append(w.m0[lo:hi:16])
            a1.append(w.m1[lo:hi:16])
            offs.append(offs[-1] + len(a0[-1]))
        sample = scene.Window(**{**w.__dict__, "m0": np.ascontiguousarray(np.concatenate(a0)),
                                 "m1": np.ascontiguousarray(np.concatenate(a1)),
                                 "pair_offsets": np.array(offs, dtype=np.uint64)})
        note = "every 16th residual block of each pair"
    reps, t0 = 0, time.perf_counter()
    while True:
        f, s, _ = orc.refine(sample, opt)
        reps += 1
        el = time.perf_counter() - t0
        if el > 10.0 or reps >= 50:
            break
    its = s.iterations * reps
    out = {"value": round(its / el, 3), "unit": "LM iterations/s", "cores": 1, "kind": "port",
           "sample": f"{reps} solve(s) of {note} ({sample.n_residual_blocks} residual blocks, "
                     f"{s.iterations} LM iterations each) in {el:.2f} s on one host core",
           "residual_blocks_per_s": round(sample.n_residual_blocks * its / el, 1)}
    # also the reference's default of 4 threads (mvo.cfg:47) and this box's CPU share
    # (OMP_NUM_THREADS on the GPU box, else the visible cores, at most 16); ~3 s each
    share = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    legs = {}
    for th in sorted({4, share}):
        if th <= 1:
            continue
        orc.set_threads(th)
        try:
            k, t1 = 0, time.perf_counter()
            while True:
                _, st, _ = orc.refine(sample, opt)
                k += 1
                e2 = time.perf_counter() - t1
                if e2 > 3.0 or k >= 50:
                    break
            legs[str(th)] = round(st.iterations * k / e2, 3)
        finally:
            orc.set_threads(1)
    out["threads"] = legs  # LM iterations/s with that many threads (evaluation and QR both parallel)
    return out


def bench_epipolar(args, rank, local_rank, world, barrier, dist, torch):
    """RANSAC baseline-direction stage (include/mvo_hip_epipolar.h <- EpipolarGeometry.cpp:34-176):
    1024 hypotheses scored against 100 000 correspondences per GPU.  The entry point takes host
    buffers; `value` counts the device time between the uploads and the read-back (HIP events), i.e.
    with the inputs resident in HBM; the PCIe-inclusive wall time is reported beside it."""
    import numpy as np
    from mvo_amd import epipolar_capi

    n, K = int(os.environ.get("MVO_BENCH_CORRESPONDENCES", "100000")), 1024
    t = epipolar_capi.synthetic_correspondences(n, seed=4000 + rank, n_samples=K)
    thr = float(np.cos(np.deg2rad(4.0)))
    ep = epipolar_capi.Epipolar(local_rank)
    for _ in range(args.warmup):
        ep.score(t["mhi"], t["mt"], t["samples"], thr)
    barrier()
    t0 = time.perf_counter()
    dev = 0.0
    for _ in range(args.steps):
        hb, hi, hp = ep.score(t["mhi"], t["mt"], t["samples"], thr)
        secs, ev = ep.last_timing()
        dev += secs
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dev], dtype=torch.float64, device=BK_DEVICE)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dev = float(tt.item())
    # pInlier: ~40 fp64 flops per (hypothesis, correspondence); the 4.8 MB of rays stay in L2
    tf = 40.0 * ev / (dev / args.steps) / 1e12
    out = {"metric": "hypothesis_correspondence_evaluations_per_sec", "value": round(ev * args.steps * world / dev, 1),
           "unit": "evaluations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dev / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"epipolar: {K} RANSAC hypotheses x {n} correspondences per GPU",
                      "parallelism": f"{world} x independent frames, no collective"},
           "ms_per_step_pcie_inclusive": round(elapsed / args.steps * 1e3, 5),
           "roofline": None,
           "roofline_compute": {"bound": "fp64-valu", "achieved": round(tf, 3), "peak": 78.6, "unit": "TFLOP/s",
                                "frac": round(tf / 78.6, 4), "flops_per_launch": int(40 * ev)}}
    if rank == 0 and world == 1 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import mvo_epipolar_oracle_py as eo

        k = 64
        t1 = time.perf_counter()
        _, hio, hpo = eo.score(t["mhi"], t["mt"], t["samples"][:k], thr)
        el = time.perf_counter() - t1
        assert np.array_equal(hio, hi[:k])
        out["cpu_baseline"] = {"value": round(k * n / el, 1), "unit": "evaluations/s", "cores": 1, "kind": "port",
                               "sample": f"the first {k} hypotheses x {n} correspondences in {el:.2f} s on one host core"}
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    ep.close()


def bench_scaler(args, rank, local_rank, world, barrier, dist, torch):
    """Closed-form scale stage (include/mvo_hip_scaler.h <- Scaler.cpp:21-68) on 10 M synthetic
    3-frame tracks per GPU, rays resident in HBM: per-track scales + radix-select median."""
    import numpy as np
    from mvo_amd import scaler_capi

    n = int(os.environ.get("MVO_BENCH_TRACKS", "10000000"))
    t = scaler_capi.synthetic_tracks(n, seed=3000 + rank)
    inp = scaler_capi.make_input(t["R0"], t["R1"], t["R2"], t["u1"], t["n1"])
    sc = scaler_capi.Scaler(local_rank)
    t0 = time.perf_counter()
    sc.upload(t["v0"], t["v1"], t["v2"])
    upload_s = time.perf_counter() - t0
    for _ in range(args.warmup):
        sc.run(inp)
    barrier()
    t0 = time.perf_counter()
    dev = tdev = 0.0
    for _ in range(args.steps):
        res, _ = sc.run(inp)
        secs, nbytes = sc.last_timing()
        tsecs, tbytes = sc.tracks_timing()
        dev += secs
        tdev += tsecs
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=BK_DEVICE)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    gbs = tbytes / (tdev / args.steps) / 1e9 if tdev > 0 else 0.0
    run_gbs = nbytes / (dev / args.steps) / 1e9
    out = {"metric": "tracks_per_sec", "value": round(n * args.steps * world / elapsed, 1), "unit": "tracks/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"scaler: {n} three-frame tracks per GPU, per-track closed-form scale + median",
                      "tracks_per_gpu": n, "parallelism": f"{world} x independent windows, no collective"},
           "median": res.median, "scale": res.scale, "upload_seconds_pcie_inclusive": round(upload_s, 6),
           # the dominant kernel: 72 B of rays in + 8 B of scale out per track
           "roofline": {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": measured_traffic("scaler", "k_scaler_tracks"),
                        "kernel": "k_scaler_tracks", "bytes_per_launch": int(tbytes),
                        "avg_launch_us": round(tdev / args.steps * 1e6, 3), "launches_timed": args.steps},
           # the whole run on the device: that kernel + the radix select (two more passes over the scales)
           "whole_run": {"achieved": round(run_gbs, 2), "unit": "GB/s", "frac": round(run_gbs / HBM_PEAK_GBS, 4),
                         "bytes": int(nbytes), "device_us": round(dev / args.steps * 1e6, 3)}}
    if rank == 0 and world == 1 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import mvo_scaler_oracle_py as so

        k = min(n, 2_000_000)
        t1 = time.perf_counter()
        so.run(t["R0"], t["R1"], t["R2"], t["u1"], t["n1"], 0.25, 2.0, t["v0"][:k], t["v1"][:k], t["v2"][:k])
        el = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(k / el, 1), "unit": "tracks/s", "cores": 1, "kind": "port",
                               "sample": f"the first {k} tracks (scales + qsort median) in {el:.2f} s on one host core"}
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    sc.close()


def bench_ba(args, rank, local_rank, world, barrier, dist, torch):
    """Explicit-landmark window BA.  ba2 = configs[1] shape (30 keyframes, 5 000 landmarks, 100 000
    observations), ba5 = the same with 20 % outliers under a Cauchy loss, ba4 = configs[3] shape
    (200 keyframes, 200 000 landmarks, 10 M observations; 28 free frames).  With N > 1 every rank
    solves its own scene (independent sequences, no collective): weak scaling."""
    from mvo_amd import ba_capi, ba_scene

    fixed = dict(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    if args.workload == "ba2":
        sc, kw = ba_scene.ba_config2(seed=2002 + rank), dict(fixed)
        desc = "ba2: 30 keyframes (28 free), 5000 landmarks x 20 frames = 100k observations, fixed 10 LM iterations"
    elif args.workload == "ba5":
        sc, kw = ba_scene.ba_config2(seed=2005 + rank, outlier_fraction=0.2), dict(loss=1, loss_scale=0.003, **fixed)
        desc = "ba5: ba2 data with 20% outlier observations, Cauchy(0.003) IRLS, fixed 10 LM iterations"
    elif args.workload == "ba4":
        sc, kw = ba_scene.ba_config4(seed=2004 + rank), dict(fixed)
        desc = "ba4: 200 keyframes (28 free), 200k landmarks x 50 frames = 10M observations, fixed 10 LM iterations"
    else:
        raise SystemExit(f"unknown workload {args.workload}")
    # ba4 with N > 1: ONE scene, landmarks range-partitioned over the ranks, the reduced camera
    # system all-reduced over RCCL before every (identical) camera solve -- strong scaling
    partitioned = world > 1 and args.workload == "ba4"
    full = sc
    if partitioned:
        from mvo_amd import sharding

        full = sc = ba_scene.ba_config4(seed=2004)
        sc, _ = sharding.shard_landmarks(full, rank, world)
    solver = ba_capi.BaSolver(local_rank)
    t0 = time.perf_counter()
    solver.upload(sc, ba_capi.make_options(**kw))
    upload_s = time.perf_counter() - t0
    if partitioned:
        buf = torch.zeros(18 * sc.M * (sc.M + 1) + 18 * sc.M + 1, dtype=torch.float64, device="cuda")  # read by the library

        def ar(op):
            def f(t):
                dist.all_reduce(t, op=op)
                torch.cuda.synchronize()
            return f

        ar_sum, ar_max = ar(dist.ReduceOp.SUM), ar(dist.ReduceOp.MAX)

        def run_solve():
            solver.solve_partitioned(ar_sum, ar_max, buf)
    else:
        run_solve = solver.solve
    for _ in range(args.warmup):
        run_solve()
    barrier()
    t0 = time.perf_counter()
    sch_s, sch_n, dev_s = 0.0, 0, 0.0
    for _ in range(args.steps):
        run_solve()
        poses, X, summ = solver.download()
        tm = solver.timing()
        sch_s += tm["schur_seconds"]
        sch_n += tm["schur_launches"]
        dev_s += tm["device_seconds"]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=BK_DEVICE)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    units = 1 if partitioned else world
    if partitioned:
        sch_n = 0  # the partitioned path is not event-timed
    its = summ.iterations
    roof = None
    if sch_n:
        dur = sch_s / sch_n
        gbs = tm["schur_bytes"] / dur / 1e9
        roof = {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": measured_traffic(args.workload, "k_ba_schur"),
                "kernel": "k_ba_schur", "bytes_per_launch": int(tm["schur_bytes"]),
                "avg_launch_us": round(dur * 1e6, 3), "launches_timed": int(sch_n)}
        # k_ba_schur is fp64-VALU bound, not HBM bound: quote it against the vector fp64 peak too
        # (AMD's published MI355X figure, 78.6 TFLOP/s; the microarchitecture guide lists none)
        tf = tm["schur_flops"] / dur / 1e12
        roof_compute = {"bound": "fp64-valu", "achieved": round(tf, 3), "peak": 78.6, "unit": "TFLOP/s",
                        "frac": round(tf / 78.6, 4), "flops_per_launch": int(tm["schur_flops"])}
    out = {
        "metric": "lm_iterations_per_sec", "value": round(its * args.steps * units / elapsed, 3),
        "unit": "LM iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True,
        "scaling": "strong" if partitioned else "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": desc, "keyframes": sc.N, "free_keyframes": sc.M, "landmarks_per_gpu": sc.n_landmarks,
                   "observations_per_gpu": sc.n_observations, "lm_iterations": its,
                   "parallelism": (f"1 scene, landmarks range-partitioned over {world} GPUs, RCCL all-reduce of the "
                                   f"reduced camera system ({18 * sc.M * (sc.M + 1) + 18 * sc.M + 1} fp64) per iteration"
                                   if partitioned else f"{world} x independent scenes, no collective")},
        "residual_jacobian_obs_per_sec": round(full.n_observations * (1 if partitioned else world) * its * args.steps / elapsed, 1),
        "device_ms_per_step": round(dev_s / args.steps * 1e3, 5),
        "upload_seconds_pcie_inclusive": round(upload_s, 6),
        "final_cost": summ.final_cost, "initial_cost": summ.initial_cost,
        "roofline": roof,
        "roofline_compute": roof_compute if sch_n else None,
    }
    if rank == 0 and world == 1 and not args.no_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import mvo_ba_oracle_py as bo

        sample, note = sc, "the whole scene"
        if sc.n_observations > 300_000:  # bound the CPU leg: the first 5000 landmarks
            import numpy as np
            from dataclasses import replace

            k = 5000
            e = int(sc.lm_offsets[k])
            sample = replace(sc, X=sc.X[:k], lm_offsets=sc.lm_offsets[:k + 1], obs_frame=sc.obs_frame[:e],
                             obs_ray=sc.obs_ray[:e], true_X=sc.true_X[:k])
            note = f"its first {k} landmarks"
        okw = dict(kw)
        okw["max_num_iterations"] = 3
        t1 = time.perf_counter()
        _, _, so, _ = bo.refine(sample, bo.make_options(**okw))
        el = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": round(so.iterations / el, 3), "unit": "LM iterations/s", "cores": 1,
                               "kind": "port",
                               "sample": f"{so.iterations} LM iterations of {note} ({sample.n_observations} observations) "
                                         f"in {el:.2f} s on one host core (self-oracle: no reference counterpart)",
                               "observations_per_s": round(sample.n_observations * so.iterations / el, 1)}
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    solver.close()


def bench_tracks(args, rank, local_rank, world, barrier, dist, torch):
    """--workload tracks2 | tracks4: the refinement of config 2 / config 4 with the feature tracks
    RESIDENT in HBM (include/mvo_hip_tracks.h).  A step is one complete
    mvo_hip_tracks_refine: count + fill kernels gather every frame pair's residual blocks on the
    device, the N frame records go up, the LM loop runs, results come back.  The tables were pushed
    before the timed region (a tracker pushes one keyframe per refine; the push is not the path).
    `roofline` is k_tracks_fill (HBM-bound scattered ray gather); the same window refined from
    host-gathered blocks through mvo_hip_refine (PCIe-inclusive) is timed beside it."""
    import numpy as np
    from mvo_amd import capi, scene, tracks_capi

    fixed10 = dict(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    if args.workload == "tracks2":
        w = scene.make_window(30, 5, 5000, 1002 + rank, span=("sliding", 20), keep_tracks=True)
        desc = "config2 from device-resident tracks: N=30 M=5, 5000 landmarks, fixed 10 LM iterations"
    elif args.workload == "tracks4":
        w = scene.make_window(200, 5, 20000, 1004 + rank, span=("anchored", 2, 200), keep_tracks=True)
        desc = "config4 (one GPU's worth) from device-resident tracks: N=200 M=5, ~9.7M residual blocks, fixed 10 LM iterations"
    else:
        raise SystemExit(f"unknown workload {args.workload}")
    opt = capi.make_options(**fixed10)
    frames = scene.track_frames(w)
    solver = capi.Solver(local_rank)
    table = tracks_capi.TrackTable(solver, max_frames=w.N, max_features=max(len(p) for _, p in frames))
    table.set_profiling(True)
    for rays, pre in frames:
        table.push(rays, pre)

    for _ in range(args.warmup):
        table.refine(w.frames, w.M, opt)
    table.refine(w.frames, w.M, opt)
    table.timing()  # the fused refine counts no residual blocks; the first timing query does (once per table state)
    barrier()
    t0 = time.perf_counter()
    fill_s = count_s = lm_dev_s = 0.0
    n_prof = 0
    for i in range(args.steps):
        timed = i % 4 == 0  # HIP events on every 4th step only, as in the main workloads
        solver.set_profiling(timed)
        table.set_profiling(timed)
        out_frames, summ = table.refine(w.frames, w.M, opt)
        if timed:
            n_prof += 1
            tt = table.timing()
            fill_s += tt["fill_seconds"]
            count_s += tt["count_seconds"]
            lm_dev_s += solver.timing()["device_seconds"]
    solver.set_profiling(True)
    table.set_profiling(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=BK_DEVICE)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the same window through the host-gathered entry point (rays cross PCIe on every call)
    host_ms = None
    if rank == 0:
        offs, m0, m1 = table.gather(w.M, w.N)
        hw = scene.Window(**{**w.__dict__, "pair_offsets": offs, "m0": m0, "m1": m1, "meta": {}})
        for _ in range(2):
            exp, es = solver.refine(hw, opt)
        k = max(3, min(args.steps, 20))
        t1 = time.perf_counter()
        for _ in range(k):
            exp, es = solver.refine(hw, opt)
        host_ms = (time.perf_counter() - t1) / k * 1e3
        # no loss: the moments are summed straight from the tables, in another order than k_fold sums the
        # gathered blocks (bit-exact with MVO_HIP_TRACKS_NO_FUSE=1, tests/test_gpu_tracks.py)
        assert np.abs(exp - out_frames).max() < 1e-8, "device-gathered and host-gathered refinement differ"

    dur = fill_s / n_prof
    gbs = tt["fill_bytes"] / dur / 1e9 if dur > 0 else 0.0
    fused = not os.environ.get("MVO_HIP_TRACKS_NO_FUSE")
    gather_kernel = "k_tracks_fold" if fused else "k_tracks_fill"
    out = {
        "metric": "lm_iterations_per_sec", "value": round(summ.iterations * args.steps * world / elapsed, 3),
        "unit": "LM iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": desc, "residual_blocks_per_gpu": int(tt["residual_blocks"]),
                   "pairs_per_window": w.n_pairs, "lm_iterations_per_window": summ.iterations,
                   "features_per_frame_max": max(len(p) for _, p in frames),
                   "parallelism": f"{world} x independent windows, no collective"},
        "gathered_blocks_per_sec": round(tt["residual_blocks"] * args.steps * world / elapsed, 1),
        "count_kernel_us": round(count_s / n_prof * 1e6, 3),
        "fill_kernel_us": round(dur * 1e6, 3),
        "solver_device_ms_per_step": round(lm_dev_s / n_prof * 1e3, 5),
        "steps_with_hip_events": n_prof,
        "host_gathered_refine_ms_pcie_inclusive": None if host_ms is None else round(host_ms, 5),
        "roofline": {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": measured_traffic(args.workload, gather_kernel),
                     "kernel": gather_kernel, "bytes_per_launch": int(tt["fill_bytes"]),
                     "avg_launch_us": round(dur * 1e6, 3), "launches_timed": n_prof},
    }
    if rank == 0 and world == 1 and not args.no_cpu:
        # CPU leg: the host-side chain walk the device gather replaces (numpy restatement, one core)
        t1 = time.perf_counter()
        reps = 0
        while True:
            scene.gather_tracks_numpy(frames, w.M)
            reps += 1
            if time.perf_counter() - t1 > 5.0 or reps >= 20:
                break
        g_s = (time.perf_counter() - t1) / reps
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import mvo_oracle_py as orc

        sample, note = hw, "the whole window"
        if hw.n_residual_blocks > 600_000:
            o2, a0, a1 = [0], [], []
            for p in range(hw.n_pairs):
                lo, hi = int(hw.pair_offsets[p]), int(hw.pair_offsets[p + 1])
                a0.append(hw.m0[lo:hi:16])
                a1.append(hw.m1[lo:hi:16])
                o2.append(o2[-1] + len(a0[-1]))
            sample = scene.Window(**{**hw.__dict__, "m0": np.ascontiguousarray(np.concatenate(a0)),
                                     "m1": np.ascontiguousarray(np.concatenate(a1)),
                                     "pair_offsets": np.array(o2, dtype=np.uint64)})
            note = "every 16th residual block of each pair, time scaled by 16"
        t1 = time.perf_counter()
        _, so, _ = orc.refine(sample, orc.make_options(**fixed10))
        s_s = (time.perf_counter() - t1) * (hw.n_residual_blocks / max(1, sample.n_residual_blocks))
        out["cpu_baseline"] = {"value": round(so.iterations / (g_s + s_s), 3), "unit": "LM iterations/s", "cores": 1,
                               "kind": "port",
                               "sample": f"numpy chain-walk gather of the whole window ({g_s * 1e3:.1f} ms, mean of {reps}) + "
                                         f"one oracle solve of {note} ({s_s:.2f} s) on one host core"}
    else:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    table.close()
    solver.close()


def self_launch(argv, gpus):
    """N > 1 without WORLD_SIZE: become the launcher.  Nothing in this process has imported torch or
    made a HIP call; the ranks are a fresh child process tree (never an exec of a process that touched
    the GPU).  Returns the child's exit code; its stdout (rank 0's one JSON line) and stderr pass through."""
    import socket
    import subprocess

    with socket.socket() as sk:  # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # RCCL across processes: dmabuf IPC only on this pool
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"bench.py: starting {gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=None,
                    help="default: config2 on one GPU, config3 (the same 4096 windows sharded) on several")
    ap.add_argument("--weak", action="store_true", help="N > 1: every rank its own copy of the workload")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-large", action="store_true", help="skip the config-4-size roofline leg")
    ap.add_argument("--no-pcie", action="store_true",
                    help="skip the host-buffer (PCIe-inclusive) legs: their fold launches -- cold behind a copy, or reading "
                         "pinned host memory in place -- carry the same kernel name as the timed region's and would be "
                         "averaged into it by a kernel trace (scripts/profile.sh)")
    args = ap.parse_args()
    global BK_DEVICE

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libmvo_hip has no CPU fallback")
    # MVO_BENCH_REHEARSE=1: every rank on GPU 0, gloo for the bench's own bookkeeping collectives -- a
    # rehearsal of the N > 1 code path (sharding, per-rank workloads, aggregation) on a one-GPU box; the
    # line it prints says so and is not a measurement of N GPUs.  RCCL needs one GPU per rank.
    rehearse = world > 1 and os.environ.get("MVO_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            BK_DEVICE = "cpu"  # the bookkeeping tensors go through gloo on the host
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from mvo_amd import capi

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    default_workload = args.workload is None
    args.workload, mode = plan(args.workload, world, args.weak)
    if args.workload == "epipolar":
        bench_epipolar(args, rank, local_rank, world, barrier, dist, torch)
        if world > 1:
            dist.destroy_process_group()
        return
    if args.workload == "scaler":
        bench_scaler(args, rank, local_rank, world, barrier, dist, torch)
        if world > 1:
            dist.destroy_process_group()
        return
    if args.workload.startswith("tracks"):
        bench_tracks(args, rank, local_rank, world, barrier, dist, torch)
        if world > 1:
            dist.destroy_process_group()
        return
    if args.workload.startswith("ba"):
        bench_ba(args, rank, local_rank, world, barrier, dist, torch)
        if world > 1:
            dist.destroy_process_group()
        return

    # config4 / config4r with N > 1: ONE window, residual blocks range-partitioned over the ranks,
    # per-pair sums all-reduced by RCCL INSIDE the library, on its stream, before every (identical)
    # LM step (mvo_hip_comm_init / mvo_hip_solve_partitioned) -- strong scaling.
    # config3 with N > 1: the same 4096 windows block-partitioned over the ranks, no collective.
    partitioned = mode == "partitioned"
    ws, opt, desc = build_workload(args.workload, rank, world, mode)
    solver = capi.Solver(local_rank)
    if partitioned:
        from mvo_amd import sharding

        full = ws[0]
        ws = [sharding.shard_observations(full, rank, world)]
        uid = [capi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)   # the unique id travels over torch's own process group
        solver.comm_init(uid[0], rank, world)
        run_solve = solver.solve_partitioned_rccl
    else:
        run_solve = solver.solve
    t0 = time.perf_counter()
    solver.upload(ws, opt)
    upload_s = time.perf_counter() - t0

    for _ in range(args.warmup):
        run_solve()
        solver.download()
    barrier()
    t0 = time.perf_counter()
    asm_s, asm_n, dev_s, first_s, host_s = 0.0, 0, 0.0, 0.0, 0.0
    # HIP events (kernel durations for `roofline`) are recorded on every 4th step of the timed
    # region only: the event records themselves cost ~8 us per solve, which is not the library's
    # default (mvo_hip_set_profiling is off unless asked for)
    every = 4
    n_prof = 0
    prof_on = None
    for i in range(args.steps):
        timed = i % every == 0
        if timed != prof_on:
            solver.set_profiling(timed)
            prof_on = timed
        run_solve()
        frames, summ = solver.download(copy=False)  # refined frames + summaries are on the host
        if timed:  # the timing record is read on the event-timed steps only (a ctypes call and a dict per step otherwise)
            tm = solver.timing()
            host_s += tm["solve_seconds"]
            n_prof += 1
            asm_s += tm["assembly_seconds"]
            asm_n += tm["assembly_launches"]
            dev_s += tm["device_seconds"]
            first_s += tm["assembly_first_seconds"]
            tm_prof = tm
    barrier()
    elapsed = time.perf_counter() - t0
    solver.set_profiling(True)
    tm = tm_prof
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=BK_DEVICE)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_windows = len(ws)
    its_per_step = sum(s.iterations for s in summ)
    blocks = sum(w.n_residual_blocks for w in ws)
    bearings = sum(w.bearing_observations for w in ws)
    units = 1 if partitioned else world   # a partitioned window is one job, not `world` jobs
    strong = mode in ("partitioned", "strong_windows")
    if world > 1 and not partitioned:   # the ranks' shares differ in size: whole-job totals
        tot = torch.tensor([float(its_per_step), float(blocks), float(bearings), float(n_windows)], dtype=torch.float64, device=BK_DEVICE)
        dist.all_reduce(tot)
        job_its, job_blocks, job_bearings, job_windows = (float(x) for x in tot.tolist())
    else:
        job_its, job_blocks, job_bearings, job_windows = float(its_per_step), float(blocks) * (world if partitioned else 1), float(bearings), float(n_windows)
    total_its = job_its * args.steps
    tm["assembly_seconds"], tm["assembly_launches"] = asm_s, asm_n
    roof = None if partitioned else roofline_of(tm, args.workload)  # the partitioned path is not event-timed
    if partitioned:
        solver.comm_destroy()

    out = {
        "metric": "lm_iterations_per_sec",
        "value": round(total_its / elapsed, 3),
        "unit": "LM iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic" if os.environ.get("MVO_BENCH_REHEARSE") != "1" or world == 1
                else "synthetic; REHEARSAL: all ranks share GPU 0 (MVO_BENCH_REHEARSE=1), not a measurement of N GPUs",
        "config": {"workload": desc, "windows_per_gpu": n_windows, "residual_blocks_per_gpu": blocks,
                   "windows_whole_job": int(job_windows), "residual_blocks_whole_job": int(job_blocks),
                   "bearing_observations_per_gpu": bearings, "pairs_per_window": ws[0].n_pairs,
                   "lm_iterations_per_window": its_per_step // n_windows,
                   "parallelism": (f"1 window range-partitioned over {world} GPUs, RCCL all-reduce of 8 fp64 per pair on the library's stream"
                                   if partitioned else
                                   f"the same {int(job_windows)} windows block-partitioned over {world} GPUs, no collective"
                                   if mode == "strong_windows" else f"{world} x independent windows, no collective")},
        # every LM iteration evaluates all residual blocks and their Jacobian rows (through the
        # per-pair moments in the no-loss case, through one observation pass otherwise)
        "residual_jacobian_obs_per_sec": round(job_blocks * (its_per_step / n_windows) * args.steps / elapsed, 1),
        "windows_per_sec": round(job_windows * args.steps / elapsed, 3),
        "device_ms_per_step": round(dev_s / n_prof * 1e3, 5),
        "host_solve_ms_per_step": round(host_s / n_prof * 1e3, 5),  # mvo_hip_solve on the event-timed steps
        "fold_kernel_us": round(first_s / n_prof * 1e6, 3),
        "steps_with_hip_events": n_prof,
        "upload_seconds_pcie_inclusive": round(upload_s, 6),
        "roofline": roof,
    }

    if rank == 0 and not partitioned and len(ws) == 1 and not args.no_pcie:
        # the call the reference would actually make: HOST buffers in and out (upload over PCIe + solve +
        # download) -- never part of `value`.  Two forms: the gather loop wrote the rays into the handle's
        # pinned staging block (mvo_hip_staging + mvo_hip_refine_staged: what the shipped adapter does), or
        # into ordinary (pageable) memory handed to mvo_hip_refine.  The staging block is filled once,
        # outside the timed calls, as the arrays of the pageable form are.
        k = 3 if blocks > 2_000_000 else 20
        solver.stage(ws[0])
        for _ in range(2):
            solver.refine_staged(ws[0], opt)
        t1 = time.perf_counter()
        for _ in range(k):
            solver.refine_staged(ws[0], opt)
        out["refine_ms_pcie_inclusive"] = round((time.perf_counter() - t1) / k * 1e3, 5)
        for _ in range(2):
            solver.refine(ws[0], opt)
        t1 = time.perf_counter()
        for _ in range(k):
            solver.refine(ws[0], opt)
        out["refine_ms_pcie_inclusive_pageable"] = round((time.perf_counter() - t1) / k * 1e3, 5)
    if strong and world > 1:
        # the efficiency baseline in the same line: the WHOLE job on rank 0's GPU alone (the other ranks
        # wait) -- all windows of config 3, or the unsplit config-4 window through the plain mvo_hip_solve
        if rank == 0:
            if partitioned:
                all_ws = [full]
            else:
                all_ws, _, _ = build_workload(args.workload, 0, 1, "single")
            solver.upload(all_ws, opt)
            solver.set_profiling(False)  # as the timed region runs three steps in four: no event records
            for _ in range(2):
                solver.solve()
            k = max(3, min(args.steps, 10))
            t1 = time.perf_counter()
            for _ in range(k):
                solver.solve()
                _, s1 = solver.download(copy=False)
            e1 = (time.perf_counter() - t1) / k
            one = sum(x.iterations for x in s1) / e1
            out["same_workload_one_gpu"] = {"value": round(one, 3), "unit": "LM iterations/s",
                                            "ms_per_step": round(e1 * 1e3, 5), "windows": len(all_ws)}
            # strong scaling: (one-GPU time / N-GPU time) / N
            out["efficiency_vs_one_gpu"] = round(out["value"] / (one * world), 4)
            del all_ws
        barrier()

    if rank == 0 and world == 1 and not args.no_large and args.workload in ("config2", "config5"):
        # the same pass kernels at a size HBM actually sees (config 4, one GPU)
        robust = args.workload == "config5"
        big_ws, big_opt, big_desc = build_workload("config4r" if robust else "config4", 0)
        solver.upload(big_ws, big_opt)
        asm_s, asm_n, k = 0.0, 0, 0
        t1 = time.perf_counter()
        for i in range(6):
            solver.solve()
            _, bs = solver.download()
            if i == 0:
                continue
            tb = solver.timing()
            asm_s += tb["assembly_seconds"]
            asm_n += tb["assembly_launches"]
            k += 1
        big_elapsed = (time.perf_counter() - t1) * k / 6
        tb["assembly_seconds"], tb["assembly_launches"] = asm_s, asm_n
        rl = roofline_of(tb, "config4r" if robust else "config4")
        rl["workload"] = big_desc
        rl["residual_blocks"] = big_ws[0].n_residual_blocks
        rl["ms_per_solve"] = round(big_elapsed / k * 1e3, 4)
        out["roofline_large"] = rl
        del big_ws

    if rank == 0 and world == 1 and not args.no_large and default_workload:
        # The driver's 1 -> 8 sequence runs `bench.py --gpus N` without a workload: N = 1 measures configs[1]
        # (above), N > 1 configs[2].  So that the sequence still holds ONE curve, this line also carries the
        # N > 1 default workload measured on this one GPU (same code path as `--workload config3`).
        base_ws, base_opt, base_desc = build_workload("config3", 0, 1, "single")
        solver.upload(base_ws, base_opt)
        solver.set_profiling(False)  # the library's default; `--workload config3` records events on every 4th step only
        for _ in range(3):
            solver.solve()
        k = 10
        t1 = time.perf_counter()
        for _ in range(k):
            solver.solve()
            _, sb = solver.download(copy=False)
        eb = (time.perf_counter() - t1) / k
        out["scaling_base"] = {"workload": base_desc, "n_gpus": 1, "value": round(sum(x.iterations for x in sb) / eb, 3),
                               "unit": "LM iterations/s", "ms_per_step": round(eb * 1e3, 5), "windows": len(base_ws)}
        del base_ws

    if rank == 0 and world == 1 and not args.no_cpu:
        kw = dict(max_num_iterations=10, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
        out["cpu_baseline"] = cpu_baseline(ws, kw, bool(opt.use_loss_function))
    else:
        out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out), flush=True)
    solver.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
