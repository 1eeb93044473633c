"""GPU parity tests: libmvo_hip (through its C ABI) against the CPU oracle and the committed
golden vectors.  Tolerances (BASELINE.json north_star): residuals and Jacobians within 1e-9
relative to the vector's infinity norm; refined baselines within 1e-6 after a fixed iteration
count; per-pair moments within 1e-12.  Nothing here reads /root/reference.
"""
import ctypes as C

import numpy as np
import pytest

from _util import FIXED10, GOLDEN_CASES, TIGHT, load_golden, refined, rel_inf
from mvo_amd import capi, scene

pytestmark = pytest.mark.gpu

TOL_RJ = 1e-9       # residual / Jacobian, relative to ||.||_inf
TOL_MOMENT = 1e-12  # per-pair S_p
TOL_FINAL = 1e-6    # refined vec / scale after a fixed iteration count


def shapes():
    return {
        "defaults_n3m2": lambda: scene.reference_defaults_window(),          # mvo.cfg:42-43
        "minimal_n2m1": lambda: scene.make_window(2, 1, 50, seed=21),
        "n6m3": lambda: scene.make_window(6, 3, 120, seed=11),
        "n6m5": lambda: scene.make_window(6, 5, 70, seed=12),                # N = M + 1
        "config1": lambda: scene.config1(),
        "ragged": lambda: scene.make_window(12, 4, 200, seed=13, span=("anchored", 2, 12)),
        "n40m5_sliding": lambda: scene.make_window(40, 5, 600, seed=14, span=("sliding", 15)),
    }


# ------------------------------------------------------------------ residual / Jacobian / moments
@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_eval_matches_golden_vectors(solver, case):
    w, z = load_golden(case)
    r, J, cost = solver.evaluate(w)
    assert rel_inf(r, z["r0"]) < TOL_RJ
    assert rel_inf(J, z["J0"]) < TOL_RJ
    assert abs(cost - float(z["cost0"])) < 1e-12 * float(z["cost0"])
    solver.upload([w])
    assert rel_inf(solver.pair_moments(), z["S"]) < TOL_MOMENT


@pytest.mark.parametrize("name", list(shapes()))
def test_eval_and_moments_match_oracle(solver, oracle, name):
    w = shapes()[name]()
    r, J, cost = solver.evaluate(w)
    ro, Jo, co = oracle.evaluate(w)
    assert rel_inf(r, ro) < TOL_RJ
    assert rel_inf(J, Jo) < TOL_RJ
    assert abs(cost - co) <= 1e-12 * co
    solver.upload([w])
    S, So = solver.pair_moments(), oracle.pair_moments(w)
    assert rel_inf(S, So) < TOL_MOMENT


# ------------------------------------------------------------------ refine(): fixed iteration count
MODES = [capi.ASSEMBLY_AUTO, capi.ASSEMBLY_PASSES, capi.ASSEMBLY_TILED, capi.ASSEMBLY_TILED | capi.ASSEMBLY_PASSES,
         # up to 256 windows the four-wavefront kernel solves every window; these two ask for the
         # one-wavefront kernel (windows of at most 64 pairs), inline and tiled
         capi.ASSEMBLY_ONE_PER_WAVE, capi.ASSEMBLY_TILED | capi.ASSEMBLY_ONE_PER_WAVE,
         capi.ASSEMBLY_PASSES | capi.ASSEMBLY_ONE_PER_WAVE]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", list(shapes()))
def test_refine_fixed_iterations_matches_oracle(solver, oracle, name, mode):
    w = shapes()[name]()
    fg, sg = solver.refine(w, capi.make_options(assembly_mode=mode, **FIXED10))
    fo, so, _ = oracle.refine(w, oracle.make_options(**FIXED10))
    assert np.abs(refined(fg, w.M) - refined(fo, w.M)).max() < TOL_FINAL
    assert abs(sg.initial_cost - so.initial_cost) <= 1e-9 * so.initial_cost
    assert abs(sg.final_cost - so.final_cost) <= 1e-8 * so.initial_cost
    # frames that are not refined are never written (IterativeRefinement.cpp:240-246)
    assert np.array_equal(fg[w.M:], w.frames[w.M:])
    assert np.array_equal(fg[:, :9], w.frames[:, :9])


@pytest.mark.parametrize("name", ["n6m3", "config1", "ragged"])
def test_iteration_trace_matches_oracle(solver, oracle, name):
    """Same accept/reject sequence, same costs and radii, iteration by iteration."""
    w = shapes()[name]()
    for mode in MODES:
        solver.set_trace(64)
        fg, sg = solver.refine(w, capi.make_options(assembly_mode=mode, **FIXED10))
        tg = solver.trace(64)
        solver.set_trace(0)
        fo, so, to = oracle.refine(w, oracle.make_options(**FIXED10))
        assert len(tg) == len(to) == 11 and sg.iterations == so.iterations == 10
        for a, b in zip(tg, to):
            assert a["step_is_successful"] == b["step_is_successful"]
            assert abs(a["cost"] - b["cost"]) <= 1e-8 * to[0]["cost"]
            assert abs(a["trust_region_radius"] - b["trust_region_radius"]) <= 1e-6 * b["trust_region_radius"]
        assert (sg.termination, sg.reason) == (so.termination, so.reason)
        assert sg.num_successful_steps == so.num_successful_steps


@pytest.mark.parametrize("tiled", [0, capi.ASSEMBLY_TILED])
@pytest.mark.parametrize("use_loss,fix", [(True, False), (False, True), (True, True)])
def test_refine_loss_and_fixed_length(solver, oracle, use_loss, fix, tiled):
    for w in (scene.config1(), scene.make_window(7, 5, 300, seed=5, outlier_fraction=0.2)):
        kw = dict(use_loss_function=use_loss, fix_length=fix, **FIXED10)
        fg, sg = solver.refine(w, capi.make_options(assembly_mode=tiled, **kw))
        fo, so, _ = oracle.refine(w, oracle.make_options(**kw))
        assert np.abs(refined(fg, w.M) - refined(fo, w.M)).max() < TOL_FINAL
        assert abs(sg.final_cost - so.final_cost) <= 1e-8 * so.initial_cost
        if fix:  # SetParameterBlockConstant(scale): lengths come back unchanged
            assert np.abs(refined(fg, w.M)[:, 3] - refined(w.frames, w.M)[:, 3]).max() < 1e-15


@pytest.mark.parametrize("case", GOLDEN_CASES)
def test_converged_minimiser_matches_scipy_golden(solver, case):
    w, z = load_golden(case)
    for mode in MODES:
        opt = capi.make_options(use_loss_function=bool(z["use_loss"]), fix_length=bool(z["fix_length"]),
                                assembly_mode=mode, **TIGHT)
        f, s = solver.refine(w, opt)
        assert abs(s.final_cost - float(z["cost_star"])) <= 1e-8 * float(z["cost_star"])
        assert np.abs(refined(f, w.M)[:, :3] - z["vec_star"]).max() < TOL_FINAL
        assert np.abs(refined(f, w.M)[:, 3] - z["scale_star"]).max() < 2e-5  # weakly observable lengths


def test_default_options_terminate_like_the_oracle(solver, oracle):
    """The reference's shipped tolerances (mvo.cfg:39-53): convergence test, not iteration cap."""
    for w in (scene.config1(), scene.make_window(7, 5, 200, seed=5), scene.reference_defaults_window(3)):
        fg, sg = solver.refine(w, capi.make_options())
        fo, so, _ = oracle.refine(w, oracle.make_options())
        assert sg.termination == so.termination == capi.OK  # CONVERGENCE
        assert abs(sg.iterations - so.iterations) <= 1
        assert abs(sg.final_cost - so.final_cost) <= 1e-6 * so.final_cost
        assert np.abs(refined(fg, w.M)[:, :3] - refined(fo, w.M)[:, :3]).max() < 1e-4


# ------------------------------------------------------------------ batches
@pytest.mark.parametrize("tiled", [0, capi.ASSEMBLY_TILED])
def test_batch_matches_single_and_oracle(solver, oracle, tiled):
    ws = [scene.config3_window(i) for i in range(48)]
    fb, sb = solver.refine_batch(ws, capi.make_options(assembly_mode=tiled, **FIXED10))
    for i in (0, 1, 17, 47):
        fo, so, _ = oracle.refine(ws[i], oracle.make_options(**FIXED10))
        assert np.abs(refined(fb[i], 5) - refined(fo, 5)).max() < TOL_FINAL
        f1, s1 = solver.refine(ws[i], capi.make_options(**FIXED10))
        assert np.abs(refined(fb[i], 5) - refined(f1, 5)).max() < 1e-9
        assert sb[i].iterations == so.iterations == 10


@pytest.mark.parametrize("tiled", [0, capi.ASSEMBLY_TILED])
def test_batch_robust_with_early_finishers(solver, oracle, tiled):
    """Windows that converge early must not disturb the others (device-side status flags)."""
    ws = [scene.make_window(6, 3, 40 + 10 * i, seed=300 + i) for i in range(12)]
    kw = dict(use_loss_function=True, max_num_iterations=30)
    fb, sb = solver.refine_batch(ws, capi.make_options(assembly_mode=tiled, **kw))
    its = set()
    for i, w in enumerate(ws):
        fo, so, _ = oracle.refine(w, oracle.make_options(**kw))
        assert sb[i].termination == so.termination
        assert abs(sb[i].iterations - so.iterations) <= 1
        assert np.abs(refined(fb[i], 3)[:, :3] - refined(fo, 3)[:, :3]).max() < 1e-4
        its.add(sb[i].iterations)
    assert len(its) > 1  # they really did stop at different iterations


# ------------------------------------------------------------------ edge cases
def test_empty_and_partially_empty_windows(solver, oracle):
    w = scene.make_window(5, 2, 30, seed=4)
    e = scene.Window(**{**w.__dict__, "m0": w.m0[:0], "m1": w.m1[:0],
                        "pair_offsets": np.zeros_like(w.pair_offsets)})
    f, s = solver.refine(e, capi.make_options())
    assert s.initial_cost == 0.0 and s.termination == capi.OK
    assert np.array_equal(refined(f, 2)[:, :3], refined(w.frames, 2)[:, :3])
    r, J, c = solver.evaluate(e)
    assert r.size == 0 and c == 0.0
    # drop the observations of every second pair
    offs = [0]
    keep0, keep1 = [], []
    for p in range(w.n_pairs):
        lo, hi = int(w.pair_offsets[p]), int(w.pair_offsets[p + 1])
        if p % 2 == 0:
            keep0.append(w.m0[lo:hi])
            keep1.append(w.m1[lo:hi])
            offs.append(offs[-1] + hi - lo)
        else:
            offs.append(offs[-1])
    h = scene.Window(**{**w.__dict__, "m0": np.concatenate(keep0), "m1": np.concatenate(keep1),
                        "pair_offsets": np.array(offs, dtype=np.uint64)})
    fg, sg = solver.refine(h, capi.make_options(**FIXED10))
    fo, so, _ = oracle.refine(h, oracle.make_options(**FIXED10))
    assert np.abs(refined(fg, 2) - refined(fo, 2)).max() < TOL_FINAL


def test_single_observation_and_odd_sizes(solver, oracle):
    for L in (1, 2, 63, 64, 65, 127, 129):
        w = scene.make_window(4, 2, L, seed=50 + L)
        r, J, _ = solver.evaluate(w)
        ro, Jo, _ = oracle.evaluate(w)
        assert rel_inf(r, ro) < TOL_RJ and rel_inf(J, Jo) < TOL_RJ
        for mode in MODES:
            kw = dict(max_num_iterations=3, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
            fg, sg = solver.refine(w, capi.make_options(assembly_mode=mode, **kw))
            fo, so, _ = oracle.refine(w, oracle.make_options(**kw))
            assert abs(sg.initial_cost - so.initial_cost) <= 1e-9 * max(so.initial_cost, 1e-300)
            if L >= 63:  # with 1-2 features the problem is rank deficient: only costs are comparable
                assert np.abs(refined(fg, 2) - refined(fo, 2)).max() < TOL_FINAL


def test_zero_iterations_and_out_of_range_lengths(solver, oracle):
    w = scene.make_window(5, 2, 30, seed=4)
    f, s = solver.refine(w, capi.make_options(max_num_iterations=0))
    assert s.iterations == 0 and s.termination == 1 and s.reason == 1
    assert np.array_equal(refined(f, 2)[:, :3], refined(w.frames, 2)[:, :3])
    # lengths outside (lowestLength, highestLength) are clamped by reverseScale (IR.cpp:350-363)
    w.frames[0, 12] = 5.0
    w.frames[1, 12] = 0.01
    fg, sg = solver.refine(w, capi.make_options(**FIXED10))
    fo, so, _ = oracle.refine(w, oracle.make_options(**FIXED10))
    assert np.abs(refined(fg, 2) - refined(fo, 2)).max() < TOL_FINAL
    assert np.all(refined(fg, 2)[:, 3] >= 0.25) and np.all(refined(fg, 2)[:, 3] <= 2.0)
    # ... and the summary says which ones, so that the adapter can log what IR.cpp:357,363 log:
    # frame 0 clamped high (bit 1), frame 1 clamped low (bit 2)
    assert sg.clamped_scales == (1 << 1) | (1 << 2)
    assert s.clamped_scales == 0
    # the exact bounds clamp too (<=, >=), infinities are refused (the reference would recurse forever)
    w.frames[0, 12], w.frames[1, 12] = 0.25, 2.0
    _, sb = solver.refine(w, capi.make_options(**FIXED10))
    assert sb.clamped_scales == (1 << 0) | (1 << 3)
    w.frames[0, 12] = np.inf
    with pytest.raises(capi.MvoHipError) as e:
        solver.refine(w, capi.make_options(**FIXED10))
    assert e.value.status == capi.ERR_INVALID_ARGUMENT
    with pytest.raises(capi.MvoHipError):   # an interval of one ulp: the clamp cannot land inside it
        solver.refine(scene.make_window(5, 2, 30, seed=4), capi.make_options(lowest_length=4.0, highest_length=np.nextafter(4.0, 5.0)))


def test_error_behaviour(solver):
    w = scene.make_window(8, 2, 10, seed=1)
    lib = capi.load()
    frames = np.ascontiguousarray(w.frames)
    pk = capi._Packed([w])
    opt = capi.make_options()
    summ = capi.Summary()

    def call(M, N, obs=pk.arr, o=opt):
        return lib.mvo_hip_refine(solver._h, frames.ctypes.data, M, N, C.byref(o), obs, C.byref(summ))

    assert call(0, 8) == capi.ERR_INVALID_ARGUMENT           # assert(numberToRefine > 0)
    assert call(8, 8) == capi.ERR_INVALID_ARGUMENT           # assert(numberToNote > numberToRefine)
    assert call(6, 8) == capi.ERR_UNSUPPORTED                # default: assert(false)
    assert call(3, 8) == capi.ERR_INVALID_ARGUMENT           # n_pairs does not match (M, N)
    wb = scene.Window(**{**w.__dict__, "pair_offsets": w.pair_offsets.copy()})
    bad = capi._Packed([wb])
    offs = bad.keep[0]
    offs[3] = offs[4] + 1                                    # non-monotone CSR
    assert call(2, 8, bad.arr) == capi.ERR_INVALID_ARGUMENT
    o2 = capi.make_options(lowest_length=2.0, highest_length=0.25)
    assert call(2, 8, pk.arr, o2) == capi.ERR_INVALID_ARGUMENT
    assert lib.mvo_hip_refine(None, frames.ctypes.data, 2, 8, C.byref(opt), pk.arr, C.byref(summ)) == capi.ERR_INVALID_ARGUMENT
    # and the handle is still usable afterwards
    f, s = solver.refine(w, opt)
    assert s.termination in (0, 1)


def test_staged_refine_equals_refine(solver):
    """mvo_hip_refine_staged (rays written by the caller into the handle's pinned staging block, which grows
    and keeps its contents in between) returns exactly what mvo_hip_refine returns on the same blocks."""
    with capi.Solver(0) as s2:
        lib = capi.load()
        fr = np.zeros((3, 13))
        opt = capi.make_options()
        assert lib.mvo_hip_refine_staged(s2._h, fr.ctypes.data, 2, 3, C.byref(opt), None) == capi.ERR_STATE  # nothing staged
        for w, kw in ((scene.reference_defaults_window(), {}), (scene.config1(), dict(use_loss_function=True, **FIXED10)),
                      (scene.config2(), FIXED10), (scene.make_window(6, 3, 120, seed=11), FIXED10)):
            o = capi.make_options(**kw)
            s2.stage(w)
            fa, sa = s2.refine_staged(w, o)
            fb, sb = solver.refine(w, o)
            assert np.array_equal(fa, fb)
            assert (sa.termination, sa.iterations, sa.final_cost) == (sb.termination, sb.iterations, sb.final_cost)
        # offsets that do not fit what was staged are refused
        s2.stage(scene.make_window(6, 3, 120, seed=11))
        assert lib.mvo_hip_refine_staged(s2._h, np.zeros((200, 13)).ctypes.data, 5, 200, C.byref(opt), None) in (
            capi.ERR_INVALID_ARGUMENT, capi.ERR_UNSUPPORTED)


def test_download_before_upload_is_an_error():
    with capi.Solver(0) as s2:
        assert capi.load().mvo_hip_solve(s2._h) == capi.ERR_STATE
        # uploaded but never solved (or a partitioned solve abandoned half-way): no stale results
        w = scene.make_window(30, 5, 400, seed=3, span=("sliding", 18))
        s2.upload([w], capi.make_options(use_loss_function=True, **FIXED10))
        with pytest.raises(capi.MvoHipError) as e:
            s2.download()
        assert e.value.status == capi.ERR_STATE
        n = C.c_int32(0)
        assert capi.load().mvo_hip_get_trace(s2._h, None, 0, C.byref(n)) == capi.ERR_STATE
        lib = capi.load()
        assert lib.mvo_hip_partial_begin(s2._h) == capi.OK
        assert lib.mvo_hip_partial_pass(s2._h) == capi.ERR_STATE   # a pass needs the point a first advance publishes
        done = C.c_int32(0)
        n_x = int(lib.mvo_hip_partial_sums_count(s2._h))
        assert n_x == capi.CLASS_DOUBLES
        import torch
        buf = torch.zeros(n_x, dtype=torch.float64, device="cuda")
        assert lib.mvo_hip_partial_sums(s2._h, buf.data_ptr(), n_x, 0) == capi.OK      # a world of one: what went out
        assert lib.mvo_hip_partial_sums(s2._h, buf.data_ptr(), n_x, 1) == capi.OK      # comes back as the total
        assert lib.mvo_hip_partial_advance(s2._h, C.byref(done)) == capi.OK and done.value == 0
        assert lib.mvo_hip_partial_pass(s2._h) == capi.OK
        with pytest.raises(capi.MvoHipError):
            s2.download()
        s2.solve()                                   # the abandoned run's done counter is not trusted
        fa, sa = s2.download()
        with capi.Solver(0) as s3:
            s3.upload([w], capi.make_options(use_loss_function=True, **FIXED10))
            s3.solve()
            fb, sb = s3.download()
        assert np.array_equal(fa, fb) and sa[0].iterations == sb[0].iterations


def test_split_phase_is_repeatable(solver):
    """solve() restarts from the uploaded initial values: two solves are bitwise identical."""
    w = scene.config1()
    solver.upload([w], capi.make_options(use_loss_function=True, assembly_mode=capi.ASSEMBLY_TILED, **FIXED10))
    solver.solve()
    f1, s1 = solver.download()
    solver.solve()
    f2, s2 = solver.download()
    assert np.array_equal(f1, f2) and s1[0].final_cost == s2[0].final_cost
    t = solver.timing()
    # 11 evaluations = the fold (which is the pass at x0 as well) + 10 passes: inside ONE persistent launch
    # (k_lm<256, 6>) for a single window of this size, as 10 k_pass launches otherwise (MVO_HIP_NO_PERSISTENT)
    assert t["residual_blocks"] == w.n_residual_blocks
    assert (t["assembly_launches"], t["lm_launches"]) in ((0, 1), (1, 1), (10, 11)), t


def test_handle_reuse_across_shapes_and_modes(oracle):
    """One handle through a sequence of different problems (window counts, sizes, assembly paths,
    the partitioned entry points in between): every result must equal a fresh handle's bit for bit --
    nothing may leak from one solve into the next (done counter, status words, staging block)."""
    import torch

    seq = [
        (lambda: [scene.config1()], dict(**FIXED10)),
        (lambda: [scene.reference_defaults_window()], dict()),
        (lambda: [scene.config3_window(i) for i in range(7)], dict(**FIXED10)),
        (lambda: [scene.make_window(30, 5, 800, seed=41, span=("sliding", 18))], dict(use_loss_function=True, **FIXED10)),
        (lambda: [scene.reference_defaults_window(seed=9)], dict(use_loss_function=True)),
        (lambda: [scene.make_window(12, 4, 200, seed=5)], dict(assembly_mode=capi.ASSEMBLY_TILED, **FIXED10)),
        (lambda: [scene.config3_window(i) for i in range(3)], dict(assembly_mode=capi.ASSEMBLY_PASSES, **FIXED10)),
    ]
    with capi.Solver(0) as shared:
        for k, (make, kw) in enumerate(seq):
            ws = make()
            opt = capi.make_options(**kw)
            if k == 3:  # a partitioned solve on the same handle in between
                shared.upload(ws, opt)
                sums = torch.zeros(shared.partial_sums_count(), dtype=torch.float64, device="cuda")
                shared.solve_partitioned(lambda t: torch.cuda.synchronize(), sums)
                part_frames, part_summ = shared.download()
            shared.upload(ws, opt)
            shared.solve()
            fa, sa = shared.download()
            shared.solve()                      # and again without a new upload
            fb, sb = shared.download()
            with capi.Solver(0) as fresh:
                fresh.upload(ws, opt)
                fresh.solve()
                fc, sc = fresh.download()
            assert np.array_equal(fa, fc) and np.array_equal(fa, fb), k
            for x, y, z in zip(sa, sb, sc):
                assert x.iterations == y.iterations == z.iterations and x.final_cost == y.final_cost == z.final_cost
            if k == 3:
                assert np.abs(np.asarray(part_frames) - np.asarray(fa)).max() < 1e-9


@pytest.mark.parametrize("shape", ["inline", "tiled"])
@pytest.mark.parametrize("use_loss", [False, True])
def test_non_finite_and_degenerate_rays_end_with_a_verdict(solver, oracle, shape, use_loss):
    """Ceres answers NaN residuals with FAILURE and the reference writes the parameters back anyway
    (IR.cpp:235-246).  Here: a NaN ray, and all rays identical (every c = 0: nothing to minimise),
    must come back promptly with a termination verdict -- no hang, no device error -- on the
    single-launch path and on the tiled passes, and the handle must work afterwards."""
    w = scene.make_window(6, 3, 60, seed=77) if shape == "inline" else scene.make_window(10, 5, 400, seed=78)
    mode = capi.ASSEMBLY_AUTO if shape == "inline" else capi.ASSEMBLY_TILED
    kw = dict(use_loss_function=use_loss, **FIXED10)
    bad = scene.Window(**{**w.__dict__, "m0": w.m0.copy(), "m1": w.m1.copy()})
    bad.m1[len(bad.m1) // 2, 1] = np.nan
    f, s = solver.refine(bad, capi.make_options(assembly_mode=mode, **kw))
    fo, so, _ = oracle.refine(bad, oracle.make_options(**kw))
    assert s.termination == so.termination == 2 and s.reason == so.reason == 8  # FAILURE: evaluation failed
    assert s.iterations == so.iterations == 0
    assert np.array_equal(f[w.M:], w.frames[w.M:])
    # untouched parameters go back (the scale through s(reverseScale(n)): equal to rounding)
    assert np.array_equal(f[:w.M, 9:12], w.frames[:w.M, 9:12]) and np.allclose(f[:w.M, 12], w.frames[:w.M, 12], rtol=1e-14)
    assert np.array_equal(f[:w.M, 9:12], fo[:w.M, 9:12])
    flat = scene.Window(**{**w.__dict__, "m0": np.tile(w.m0[:1], (len(w.m0), 1)), "m1": np.tile(w.m0[:1], (len(w.m1), 1))})
    f, s = solver.refine(flat, capi.make_options(assembly_mode=mode, **kw))
    assert s.termination in (0, 1, 2) and s.iterations <= 10
    good, sg = solver.refine(w, capi.make_options(assembly_mode=mode, **kw))
    go, so, _ = oracle.refine(w, oracle.make_options(**kw))
    assert np.abs(refined(good, w.M) - refined(go, w.M)).max() < TOL_FINAL


def _rank_one_window(v0, N, M, n=8):
    """Identity rotations, every residual block the same pair of rays: c = (1, 0, 0) for all of them,
    r = u_x.  LM drives u_x to 1e-160 and the cost underflows to exactly 0; from there every step has
    model_cost_change = 0 -- five invalid steps in a row, Ceres' FAILURE, after ~20 accepted steps."""
    frames = np.zeros((N, 13))
    frames[:, 0] = frames[:, 4] = frames[:, 8] = 1.0
    frames[:, 9:12] = np.asarray(v0) / np.linalg.norm(v0)
    frames[:, 12] = 1.0
    P = scene.num_pairs(M, N)
    return scene.Window(N=N, M=M, frames=frames, pair_offsets=np.arange(P + 1, dtype=np.uint64) * n,
                        m0=np.tile([0.0, 1.0, 0.0], (n * P, 1)), m1=np.tile([0.0, 0.0, 1.0], (n * P, 1)),
                        true_vec=np.zeros((N, 3)), true_scale=np.zeros(N))


@pytest.mark.parametrize("mode", [capi.ASSEMBLY_AUTO, capi.ASSEMBLY_TILED])
def test_failure_after_progress_writes_the_initial_values_back(solver, oracle, mode):
    """Ceres copies the solution into the user's arrays only when the summary is usable; after FAILURE
    they keep their initial values and IR.cpp:240-246 writes vec_0 and s(reverseScale(scale_0)) --
    not the lowest-cost iterate the solve had reached before it failed.  Whether a given window ends
    in FAILURE is decided among denormals (the two sides form model_cost_change differently), so a
    family of windows is run: most must fail on both sides, and every failure must hand the inputs back."""
    kw = dict(max_num_iterations=60, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    rng = np.random.default_rng(5)
    both = 0
    for i in range(16):
        v0 = [rng.uniform(0.05, 0.6), rng.uniform(0.05, 0.6), 0.8]
        N, M = [(2, 1), (3, 2)][i % 2]
        w = _rank_one_window(v0, N, M, n=[8, 1, 3, 50][i % 4])
        fo, so, _ = oracle.refine(w, oracle.make_options(**kw))
        fg, sg = solver.refine(w, capi.make_options(assembly_mode=mode, **kw))
        for f, sm in ((fg, sg), (fo, so)):
            if sm.termination == 2:
                assert sm.reason == 6 and sm.num_successful_steps > 5 and sm.final_cost < 1e-300  # progress was made ...
                assert np.array_equal(f[:M, 9:12], w.frames[:M, 9:12])                            # ... and is not handed back
                assert np.allclose(f[:M, 12], w.frames[:M, 12], rtol=1e-14, atol=0)
        if sg.termination == 2 and so.termination == 2:
            both += 1
            assert sg.iterations == so.iterations and np.array_equal(fg[:M, 9:12], fo[:M, 9:12])
    assert both >= 10


# ------------------------------------------------------------------ the saturated branch of scaleTemplated
@pytest.mark.parametrize("name", ["n6m3", "config1"])
def test_saturated_sigmoid_branch_on_the_device(solver, oracle, name):
    """IR.cpp:301-307: exp(-t) infinite => s = lowestLength as a constant (derivative 0).  reverseScale
    never yields such a t (|t| <~ 36), so residuals / Jacobian are evaluated, and the LM loop is
    started, at an explicit point with t = -800 (and one frame at t = +800, the other flat end)."""
    w = shapes()[name]()
    M = w.M
    pt = np.zeros((M, 4))
    pt[:, :3] = w.frames[:M, 9:12]
    pt[:, 3] = [capi.reverse_scale(s) for s in w.frames[:M, 12]]
    pt[0, 3] = -800.0
    if M > 2:
        pt[2, 3] = 800.0
    assert capi.scale(-800.0) == 0.25 and capi.scale(800.0) == 2.0
    r, J, c = solver.evaluate(w, point=pt)
    ro, Jo, co = oracle.evaluate_at(w, pt)
    assert rel_inf(r, ro) < TOL_RJ and rel_inf(J, Jo) < TOL_RJ and abs(c - co) <= 1e-12 * co
    assert np.all(J[:, 2] == 0.0) and np.all(Jo[:, 2] == 0.0)       # d r / d t_0 = 0 on the saturated branch
    assert np.abs(J[:, 5]).max() > 0                                 # ... and only there
    r0, _, _ = solver.evaluate(w)                                    # and the point really was used
    assert rel_inf(r, r0) > 1e-3
    for mode in MODES:
        for kw in (dict(**FIXED10), dict(use_loss_function=True, **FIXED10)):
            solver.upload([w], capi.make_options(assembly_mode=mode, **kw))
            solver.set_point(0, pt)
            solver.solve()
            fg, sg = solver.download()
            fo, so, _ = oracle.refine(w, oracle.make_options(**kw), point=pt)
            # with tolerances 0 a converged solve stops at the first cost change of exactly 0.0: that
            # iteration is rounding dependent, so verdicts are compared where both runs went to the cap
            if sg[0].reason == 1 and so.reason == 1:
                assert (sg[0].termination, sg[0].iterations) == (so.termination, so.iterations)
            assert sg[0].iterations >= 6 and so.iterations >= 6
            assert abs(sg[0].initial_cost - so.initial_cost) <= 1e-9 * so.initial_cost
            assert abs(sg[0].final_cost - so.final_cost) <= 1e-8 * so.initial_cost
            assert np.abs(refined(fg[0], M) - refined(fo, M)).max() < TOL_FINAL
            assert fg[0][0, 12] == 0.25                              # t_0 never leaves the flat branch
