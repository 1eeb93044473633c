"""GPU parity of the device-resident track tables (include/mvo_hip_tracks.h) through the C ABI.

The on-device gather replaces the host-side chain walk of IterativeRefinement::refine
(/root/reference/mvo/src/algorithms/IterativeRefinement.cpp:116-165, sliding_window/Frame.hpp:258-302,
operations/FeatureOperations.cpp:117-122).  It is index and copy work plus one normalisation, so the
bar is BIT-EXACT against the numpy restatement scene.gather_tracks_numpy (offsets, block order and
ray values), and mvo_hip_tracks_refine must return exactly what mvo_hip_refine returns on the
host-gathered blocks (same kernels, same inputs: array_equal)."""
import numpy as np
import pytest

from mvo_amd import capi, scaler_capi, scene, tracks_capi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def solver():
    s = capi.Solver(0)
    yield s
    s.close()


def _push_all(table, frames):
    for rays, pre in frames:
        table.push(rays, pre)


CASES = [
    # N, M, landmarks, span
    (3, 1, 40, "full"),
    (10, 5, 300, "full"),                  # BASELINE.json configs[0]
    (8, 3, 500, ("sliding", 4)),
    (12, 5, 2000, ("anchored", 2, 12)),
    (30, 5, 5000, ("sliding", 8)),        # the shape of configs[1], fewer landmarks
    (6, 2, 1, "full"),                     # a single track
    (20, 4, 777, ("anchored", 1, 3)),     # many tracks too short for any pair
]


@pytest.mark.parametrize("N,M,L,span", CASES, ids=lambda v: str(v).replace(" ", ""))
def test_gather_bit_exact(solver, N, M, L, span):
    w = scene.make_window(N, M, L, seed=4100 + N, span=span, keep_tracks=True)
    frames = scene.track_frames(w)
    width = max(len(p) for _, p in frames)
    with tracks_capi.TrackTable(solver, max_frames=N, max_features=max(width, 1)) as table:
        _push_all(table, frames)
        offs, m0, m1 = table.gather(M, N)
    eo, e0, e1 = scene.gather_tracks_numpy(frames, M)
    assert np.array_equal(offs, eo)
    assert np.array_equal(offs, w.pair_offsets)  # the window builder's own count
    assert np.array_equal(m0, e0) and np.array_equal(m1, e1)
    if len(e0):  # and they are the window's unit rays to rounding (those were normalised before /z)
        assert np.abs(m0 - w.m0).max() <= 4e-16 and np.abs(m1 - w.m1).max() <= 4e-16


@pytest.mark.parametrize("use_loss", [False, True])
@pytest.mark.parametrize("N,M,L,span", [(10, 5, 300, "full"), (30, 5, 5000, ("sliding", 8)), (8, 3, 500, ("sliding", 4))],
                         ids=["config1", "config2-shape", "small"])
def test_refine_equals_host_gathered_refine(solver, N, M, L, span, use_loss):
    w = scene.make_window(N, M, L, seed=4200 + N, span=span, keep_tracks=True)
    frames = scene.track_frames(w)
    opt = capi.make_options(max_num_iterations=15, function_tolerance=0.0, gradient_tolerance=0.0,
                            parameter_tolerance=0.0, use_loss_function=use_loss)
    with tracks_capi.TrackTable(solver, max_frames=N + 3, max_features=max(len(p) for _, p in frames)) as table:
        _push_all(table, frames)
        got, gs = table.refine(w.frames, M, opt)
        offs, m0, m1 = table.gather(M, N)
    w2 = scene.Window(N=N, M=M, frames=w.frames, pair_offsets=offs, m0=m0, m1=m1, true_vec=w.true_vec,
                      true_scale=w.true_scale, bearing_observations=w.bearing_observations, meta={})
    exp, es = solver.refine(w2, opt)
    if use_loss:
        # the blocks are gathered into the solver's ray buffers: same kernels, same inputs
        assert np.array_equal(got, exp)
        assert gs.iterations == es.iterations and gs.final_cost == es.final_cost and gs.termination == es.termination
    else:
        # no loss: the per-pair moments are summed straight from the tables (k_tracks_fold), in another
        # order than k_fold sums the gathered blocks -- rounding apart, the same solve
        assert gs.iterations == es.iterations and gs.termination == es.termination
        assert abs(gs.initial_cost - es.initial_cost) <= 1e-12 * es.initial_cost
        assert abs(gs.final_cost - es.final_cost) <= 1e-9 * es.initial_cost
        assert np.abs(got - exp).max() < 1e-8
        # ... and with the fusion switched off it is the very same solve again
        import os
        os.environ["MVO_HIP_TRACKS_NO_FUSE"] = "1"
        try:
            with tracks_capi.TrackTable(solver, max_frames=N + 3, max_features=max(len(p) for _, p in frames)) as table:
                _push_all(table, frames)
                got2, gs2 = table.refine(w.frames, M, opt)
        finally:
            del os.environ["MVO_HIP_TRACKS_NO_FUSE"]
        assert np.array_equal(got2, exp)
        assert gs2.iterations == es.iterations and gs2.final_cost == es.final_cost
    # and the refinement did its job on the clean scene
    assert gs.final_cost < gs.initial_cost


@pytest.mark.parametrize("N,M,L,span", CASES, ids=lambda v: str(v).replace(" ", ""))
def test_fused_fold_moments(solver, N, M, L, span):
    """S_p summed straight from the track tables (k_tracks_fold + k_tracks_reduce) against the moments
    k_fold forms from the host-gathered blocks: 1e-12 of the largest entry (the order of summation
    differs: feature groups here, chunks of consecutive blocks there)."""
    w = scene.make_window(N, M, L, seed=4100 + N, span=span, keep_tracks=True)
    frames = scene.track_frames(w)
    width = max(len(p) for _, p in frames)
    with tracks_capi.TrackTable(solver, max_frames=N, max_features=max(width, 1)) as table:
        _push_all(table, frames)
        S = table.pair_moments(w.frames, M)
        offs, m0, m1 = table.gather(M, N)
    w2 = scene.Window(N=N, M=M, frames=w.frames, pair_offsets=offs, m0=m0, m1=m1, true_vec=w.true_vec,
                      true_scale=w.true_scale, bearing_observations=w.bearing_observations, meta={})
    solver.upload([w2])
    So = solver.pair_moments()
    assert S.shape == So.shape
    assert np.abs(S - So).max() <= 1e-12 * max(np.abs(So).max(), 1e-300)
    empty = np.diff(offs.astype(np.int64)) == 0
    assert np.all(S[empty] == 0.0)


def test_ring_slides_over_a_sequence(solver):
    """A 14-frame sequence through an 8-slot ring: after every push beyond the 6th, the newest 6
    keyframes must gather exactly as a fresh table holding only those 6 would (older slots are
    overwritten; counters that reach beyond the ring are still right)."""
    w = scene.make_window(14, 3, 400, seed=4301, span=("sliding", 9), keep_tracks=True)
    frames = scene.track_frames(w)  # oldest first
    width = max(len(p) for _, p in frames)
    N, M = 6, 3
    with tracks_capi.TrackTable(solver, max_frames=8, max_features=width) as ring:
        for k, (rays, pre) in enumerate(frames):
            ring.push(rays, pre)
            if k + 1 < N:
                with pytest.raises(capi.MvoHipError) as e:
                    ring.gather(M, N)
                assert e.value.status == capi.ERR_STATE
                continue
            offs, m0, m1 = ring.gather(M, N)
            window = list(frames[k + 1 - N:k + 1])
            # the oldest frame of the sub-window keeps its predecessor indices; the numpy
            # restatement ignores them for the last frame exactly as the chain walk never follows them
            eo, e0, e1 = scene.gather_tracks_numpy(window, M)
            # counters in the ring count frames beyond the window too; only >= past matters
            assert np.array_equal(offs, eo), k
            assert np.array_equal(m0, e0) and np.array_equal(m1, e1), k
        ring.reset()
        with pytest.raises(capi.MvoHipError):
            ring.gather(M, N)


def test_invalid_predecessors_start_new_tracks(solver):
    """Frame.cpp:138-144: a negative (or out-of-range) predecessor index starts a new track."""
    rng = np.random.default_rng(5)
    n = 300
    f_old = (rng.normal(size=(n, 3)) + [0, 0, 4.0], -np.ones(n, dtype=np.int32))
    pre = np.arange(n, dtype=np.int32)
    pre[::3] = -1
    pre[1::7] = n + 5  # out of range: treated as "no predecessor"
    f_new = (rng.normal(size=(n, 3)) + [0, 0, 4.0], pre)
    with tracks_capi.TrackTable(solver, max_frames=2, max_features=n) as table:
        _push_all(table, [f_old, f_new])
        offs, m0, m1 = table.gather(1, 2)
    eo, e0, e1 = scene.gather_tracks_numpy([f_old, f_new], 1)
    keep = (pre >= 0) & (pre < n)
    assert int(offs[-1]) == int(keep.sum()) == int(eo[-1])
    assert np.array_equal(m0, e0) and np.array_equal(m1, e1)


def test_append_to_newest_keyframe(solver):
    """Frame::mergeFrame (Frame.cpp:103-119) appends features without predecessor to a keyframe that
    is already in the window; the next keyframe may continue them.  The table after push / append /
    push must gather exactly as one that received the grown frames in one piece."""
    rng = np.random.default_rng(11)
    ray = lambda n: rng.normal(size=(n, 3)) * 0.2 + [0, 0, 1.0]
    f0 = (ray(100), -np.ones(100, dtype=np.int32))
    f1a = (ray(90), np.arange(90, dtype=np.int32))           # 90 of the 100 continue
    extra = ray(37)                                            # appended to frame 1 later
    f1 = (np.concatenate([f1a[0], extra]), np.concatenate([f1a[1], -np.ones(37, dtype=np.int32)]))
    pre2 = np.concatenate([np.arange(0, 90, 2), np.arange(90, 127)]).astype(np.int32)  # old and appended tracks
    f2 = (ray(len(pre2)), pre2)
    f3 = (ray(len(pre2)), np.arange(len(pre2), dtype=np.int32))
    with tracks_capi.TrackTable(solver, max_frames=4, max_features=128) as table:
        table.push(*f0)
        table.push(*f1a)
        table.append(extra)
        with pytest.raises(capi.MvoHipError) as e:
            table.append(ray(2))  # 127 + 2 > 128
        assert e.value.status == capi.ERR_UNSUPPORTED
        table.push(*f2)
        table.push(*f3)
        offs, m0, m1 = table.gather(3, 4)
    eo, e0, e1 = scene.gather_tracks_numpy([f0, f1, f2, f3], 3)
    assert int(eo[-1]) > 0 and np.array_equal(offs, eo)
    assert np.array_equal(m0, e0) and np.array_equal(m1, e1)


def test_argument_errors(solver):
    with pytest.raises(capi.MvoHipError) as e:
        tracks_capi.TrackTable(solver, max_frames=1, max_features=10)
    assert e.value.status == capi.ERR_INVALID_ARGUMENT
    with tracks_capi.TrackTable(solver, max_frames=4, max_features=10) as table:
        with pytest.raises(capi.MvoHipError) as e:
            table.push(np.ones((11, 3)), np.zeros(11, dtype=np.int32))
        assert e.value.status == capi.ERR_UNSUPPORTED
        table.push(np.ones((10, 3)), -np.ones(10, dtype=np.int32))
        table.push(np.ones((10, 3)), np.arange(10, dtype=np.int32))
        for M, N, status in [(0, 2, capi.ERR_INVALID_ARGUMENT), (2, 2, capi.ERR_INVALID_ARGUMENT),
                             (6, 7, capi.ERR_UNSUPPORTED), (1, 3, capi.ERR_STATE)]:
            with pytest.raises(capi.MvoHipError) as e:
                table.gather(M, N)
            assert e.value.status == status
        offs, m0, m1 = table.gather(1, 2)
        assert int(offs[-1]) == 10
        # the refine entry point reports the same conditions, and NULL arguments
        frames = scene.make_window(3, 1, 5, seed=2).frames
        with pytest.raises(capi.MvoHipError) as e:
            table.refine(frames, 1)  # three keyframes asked for, two pushed
        assert e.value.status == capi.ERR_STATE
        lib = tracks_capi.load()
        assert lib.mvo_hip_tracks_refine(table._h, None, 1, 2, None, None) == capi.ERR_INVALID_ARGUMENT
        assert lib.mvo_hip_tracks_push(table._h, 3, None, None) == capi.ERR_INVALID_ARGUMENT
        assert lib.mvo_hip_tracks_gather(table._h, 1, 2, None, None, None, 0) == capi.ERR_INVALID_ARGUMENT
        assert lib.mvo_hip_tracks_push(None, 0, None, None) == capi.ERR_INVALID_ARGUMENT
        too_small = np.zeros(3, dtype=np.uint64)
        m = np.zeros((4, 3))
        assert lib.mvo_hip_tracks_gather(table._h, 1, 2, too_small.ctypes.data, m.ctypes.data, m.ctypes.data, 4) \
            == capi.ERR_INVALID_ARGUMENT  # 10 blocks do not fit a capacity of 4
        tm = table.timing()
        assert tm["residual_blocks"] == 10 and tm["fill_seconds"] == 0.0  # profiling is off by default


def test_refine_on_empty_gather(solver):
    """No track spans two frames: zero residual blocks, the solver reports what mvo_hip_refine
    reports for an empty window (frames unchanged)."""
    N, M = 4, 2
    with tracks_capi.TrackTable(solver, max_frames=N, max_features=16) as table:
        for _ in range(N):
            table.push(np.tile([0.1, 0.2, 1.0], (16, 1)), -np.ones(16, dtype=np.int32))
        offs, m0, m1 = table.gather(M, N)
        assert int(offs[-1]) == 0 and len(m0) == 0
        w = scene.make_window(N, M, 10, seed=1, span="full")
        got, gs = table.refine(w.frames, M)
    empty = scene.Window(N=N, M=M, frames=w.frames, pair_offsets=offs, m0=m0, m1=m1, true_vec=w.true_vec,
                         true_scale=w.true_scale, bearing_observations=0, meta={})
    exp, es = solver.refine(empty)
    assert np.array_equal(got, exp) and gs.termination == es.termination


@pytest.mark.parametrize("N,L,span,age", [(3, 50, "full", 0), (6, 3000, ("sliding", 4), 0), (6, 3000, ("sliding", 4), 2),
                                          (5, 40000, ("anchored", 1, 5), 1)], ids=["ref-size", "sliding", "older", "large"])
def test_scale_stage_from_the_tables(solver, N, L, span, age):
    """mvo_hip_tracks_scale: Scaler::stage's three-frame gather (Scaler.cpp:22, Frame.hpp:103-152) on the
    device + the scale stage.  Against the host route -- the same three-frame tracks gathered in numpy
    (features of the frame with counter >= 2, in its feature order, two predecessor links) from the
    table's own unit rays and handed to mvo_hip_scaler_scale: identical inputs, so per-track scales
    and median must be bit-identical."""
    w = scene.make_window(N, age + 1, L, seed=4500 + N + age, span=span, keep_tracks=True)
    frames = scene.track_frames(w)  # oldest first
    fr = frames[::-1]  # index 0 = newest
    unit = []
    for rays, _ in fr:
        sc = 1.0 / np.sqrt((rays[:, 0] * rays[:, 0] + rays[:, 1] * rays[:, 1]) + rays[:, 2] * rays[:, 2])
        unit.append(rays * sc[:, None])
    pre0, pre1 = fr[age][1], fr[age + 1][1]
    ok0 = pre0 >= 0
    ok1 = np.zeros(len(pre0), dtype=bool)
    ok1[ok0] = pre1[pre0[ok0]] >= 0
    sel = np.nonzero(ok1)[0]  # counter >= 2
    i1 = pre0[sel]
    i2 = pre1[i1]
    v0, v1, v2 = unit[age][sel], unit[age + 1][i1], unit[age + 2][i2]
    t = scaler_capi.synthetic_tracks(4, seed=9)  # rotations / baseline of the input record; the rays come from the table
    inp = scaler_capi.make_input(t["R0"], t["R1"], t["R2"], t["u1"], t["n1"], 0.25, 2.0)
    with scaler_capi.Scaler(0) as sc_dev, scaler_capi.Scaler(0) as sc_host:
        with tracks_capi.TrackTable(solver, max_frames=N + 1, max_features=max(len(p) for _, p in frames)) as table:
            _push_all(table, frames)
            res, per = table.scale(sc_dev, inp, age=age, per_track=True)
            with pytest.raises(capi.MvoHipError) as e:
                table.scale(sc_dev, inp, age=N - 2)  # needs two older keyframes
            assert e.value.status == capi.ERR_STATE
        ref, per_ref = sc_host.scale(inp, np.ascontiguousarray(v0), np.ascontiguousarray(v1), np.ascontiguousarray(v2))
    assert res.n_tracks == len(sel) == ref.n_tracks and len(sel) > 0
    assert np.array_equal(per, per_ref)
    assert res.median == ref.median and res.scale == ref.scale and res.clamped == ref.clamped


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_gather_with_shuffled_feature_order(solver, seed):
    """A tracker does not keep features in landmark order: every frame lists the survivors of the
    previous one in RANDOM order, interleaved with new features.  Chains then hop all over the
    tables; offsets, block order and rays must still equal the numpy chain walk bit for bit, and
    the scale stage's three-frame gather its numpy counterpart."""
    rng = np.random.default_rng(900 + seed)
    N, M = 9, 4
    frames = []
    n_prev = 0
    for k in range(N):  # oldest first
        survivors = rng.permutation(n_prev)[: int(n_prev * rng.uniform(0.6, 0.95))]
        n_new = int(rng.integers(50, 400))
        pre = np.concatenate([survivors, -np.ones(n_new, dtype=np.int64)]).astype(np.int32)
        rng.shuffle(pre)
        rays = np.column_stack([rng.uniform(-0.6, 0.6, len(pre)), rng.uniform(-0.4, 0.4, len(pre)), np.ones(len(pre))])
        frames.append((np.ascontiguousarray(rays), pre))
        n_prev = len(pre)
    width = max(len(p) for _, p in frames)
    t = scaler_capi.synthetic_tracks(4, seed=3)
    inp = scaler_capi.make_input(t["R0"], t["R1"], t["R2"], t["u1"], t["n1"], 0.25, 2.0)
    with tracks_capi.TrackTable(solver, max_frames=N, max_features=width) as table, scaler_capi.Scaler(0) as sc:
        _push_all(table, frames)
        offs, m0, m1 = table.gather(M, N)
        res, per = table.scale(sc, inp, age=1, per_track=True)
        eo, e0, e1 = scene.gather_tracks_numpy(frames, M)
        assert int(eo[-1]) > 1000 and np.array_equal(offs, eo)
        assert np.array_equal(m0, e0) and np.array_equal(m1, e1)
        fr = frames[::-1]
        unit = []  # the table normalises as v * (1 / |v|): restate exactly
        for r, _ in fr:
            scl = 1.0 / np.sqrt((r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1]) + r[:, 2] * r[:, 2])
            unit.append(r * scl[:, None])
        p0, p1 = fr[1][1], fr[2][1]
        sel = np.array([i for i in range(len(p0)) if p0[i] >= 0 and p1[p0[i]] >= 0], dtype=np.int64)
        ref, per_ref = sc.scale(inp, np.ascontiguousarray(unit[1][sel]), np.ascontiguousarray(unit[2][p0[sel]]),
                                np.ascontiguousarray(unit[3][p1[p0[sel]]]))
        assert res.n_tracks == len(sel) > 100 and np.array_equal(per, per_ref) and res.median == ref.median


LONG_CASES = [
    # windows long enough for the skip links of 16 / 32 / 64 / 128 frames (k_tracks_skip) to be followed
    (40, 5, 300, ("anchored", 2, 40)),
    (80, 5, 400, ("sliding", 50)),
    (150, 5, 150, "full"),
    (200, 4, 300, ("anchored", 2, 200)),   # the shape of configs[3], fewer landmarks
]


@pytest.mark.parametrize("N,M,L,span", LONG_CASES, ids=lambda v: str(v).replace(" ", ""))
def test_long_windows_follow_skip_links(solver, N, M, L, span):
    """Walker wavefronts take 16 steps each and start from a skip link; the gather must stay bit-exact
    against the numpy restatement, and gather and fused moments must be the very bytes the single
    whole-chain walker (MVO_HIP_TRACKS_NO_SKIP) produces."""
    import os
    w = scene.make_window(N, M, L, seed=4700 + N, span=span, keep_tracks=True)
    frames = scene.track_frames(w)
    width = max(len(p) for _, p in frames)
    with tracks_capi.TrackTable(solver, max_frames=N, max_features=max(width, 1)) as table:
        _push_all(table, frames)
        offs, m0, m1 = table.gather(M, N)
        S = table.pair_moments(w.frames, M)
        os.environ["MVO_HIP_TRACKS_NO_SKIP"] = "1"
        try:
            offs1, m01, m11 = table.gather(M, N)
            S1 = table.pair_moments(w.frames, M)
        finally:
            del os.environ["MVO_HIP_TRACKS_NO_SKIP"]
    eo, e0, e1 = scene.gather_tracks_numpy(frames, M)
    assert np.array_equal(offs, eo) and np.array_equal(offs1, eo)
    assert np.array_equal(m0, e0) and np.array_equal(m1, e1)
    assert np.array_equal(m01, e0) and np.array_equal(m11, e1)
    assert np.array_equal(S, S1)
    assert int(offs[-1]) > 0


def test_skip_links_in_a_sliding_ring(solver):
    """90 keyframes through a 70-slot ring, windows of 60: skip links are written at push relative to the
    frame pushed, must survive the ring wrapping round, and links that would reach beyond the ring are
    never followed.  A few features are appended to every third keyframe (new tracks: no links)."""
    w = scene.make_window(90, 3, 250, seed=4777, span=("sliding", 70), keep_tracks=True)
    frames = scene.track_frames(w)  # oldest first
    width = max(len(p) for _, p in frames) + 8
    N, M = 60, 3
    rng = np.random.default_rng(5)
    kept = []
    with tracks_capi.TrackTable(solver, max_frames=70, max_features=width) as ring:
        for k, (rays, pre) in enumerate(frames):
            ring.push(rays, pre)
            if k % 3 == 0:
                extra = rng.normal(size=(5, 3)) + np.array([0.0, 0.0, 4.0])
                ring.append(extra)
                rays = np.concatenate([rays, extra])
                pre = np.concatenate([pre, -np.ones(5, dtype=pre.dtype)])
            kept.append((rays, pre))
            if k + 1 >= N and (k % 7 == 0 or k + 1 == len(frames)):
                offs, m0, m1 = ring.gather(M, N)
                eo, e0, e1 = scene.gather_tracks_numpy(kept[k + 1 - N:k + 1], M)
                assert np.array_equal(offs, eo), k
                assert np.array_equal(m0, e0) and np.array_equal(m1, e1), k


def test_refine_sequence_on_a_sliding_ring_with_changing_shapes(solver):
    """The fused path queues its fold before the host knows the pair counts and reuses the partial-sum
    and count tables from call to call: a sequence of refines with changing (M, N) on a ring that keeps
    sliding must give, every time, what mvo_hip_refine gives on the blocks gathered at that moment."""
    w = scene.make_window(40, 5, 350, seed=4888, span=("sliding", 25), keep_tracks=True)
    frames = scene.track_frames(w)  # oldest first
    width = max(len(p) for _, p in frames)
    opt = capi.make_options(max_num_iterations=6, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
    shapes = [(3, 8), (5, 30), (2, 5), (5, 22), (1, 2), (4, 30)]
    with tracks_capi.TrackTable(solver, max_frames=32, max_features=width) as ring:
        for k, (rays, pre) in enumerate(frames):
            ring.push(rays, pre)
            M, N = shapes[k % len(shapes)]
            if k + 1 < N:
                continue
            # newest-first frame records of the N newest keyframes (any consistent poses do: reuse the window's)
            fr = w.frames[w.N - 1 - k: w.N - 1 - k + N].copy()
            got, gs = ring.refine(fr, M, opt)
            offs, m0, m1 = ring.gather(M, N)
            w2 = scene.Window(N=N, M=M, frames=fr, pair_offsets=offs, m0=m0, m1=m1, true_vec=w.true_vec[:N],
                              true_scale=w.true_scale[:N], bearing_observations=0, meta={})
            exp, es = solver.refine(w2, opt)
            assert gs.iterations == es.iterations and gs.termination == es.termination, k
            assert abs(gs.initial_cost - es.initial_cost) <= 1e-12 * max(es.initial_cost, 1e-300), k
            # the two routes differ by the summation order of S_p (1e-12); a window with a gauge direction
            # (k = 14: M = 2, N = 5) amplifies that to 1e-9 ... 1e-7 depending on the last bits of the
            # trajectory -- the bar is the library's parity bar for refined parameters, 1e-6
            assert np.abs(got - exp).max() < 1e-6, k


def test_fused_refine_reports_its_blocks_when_asked(solver):
    """The fused refine (no loss) neither counts its residual blocks nor waits for the device before the
    solve is queued; mvo_hip_tracks_get_timing counts them on demand -- the number the gather reports --
    and again after the table has changed."""
    w = scene.make_window(12, 4, 300, seed=4901, span=("sliding", 7), keep_tracks=True)
    frames = scene.track_frames(w)
    opt = capi.make_options(max_num_iterations=3)
    with tracks_capi.TrackTable(solver, max_frames=16, max_features=max(len(p) for _, p in frames)) as table:
        _push_all(table, frames[:-1])
        table.refine(w.frames[1:], 4, opt)                    # the 11 keyframes pushed so far
        offs, _, _ = table.gather(4, 11)
        table.refine(w.frames[1:], 4, opt)
        tt = table.timing()
        assert tt["residual_blocks"] == int(offs[-1]) > 0 and tt["count_seconds"] == 0.0
        assert tt["fill_bytes"] > 28 * tt["residual_blocks"]
        table.push(*frames[-1])
        table.refine(w.frames, 4, opt)
        offs2, _, _ = table.gather(4, 12)
        table.refine(w.frames, 4, opt)
        assert table.timing()["residual_blocks"] == int(offs2[-1]) == int(w.pair_offsets[-1])
