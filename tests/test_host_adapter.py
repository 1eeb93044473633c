"""The reference-side C++ adapter (host/IterativeRefinementHip.hpp) driven through stand-ins of the
reference's Frame/Feature track structure: its gathering must reproduce the CSR buffers of the
python generator (same pair order, same track selection rule as Frame::getCorrespondingFeatures,
Frame.hpp:258-302), and its refine() must return what the oracle returns."""
import os
import struct
import subprocess

import numpy as np
import pytest

from _util import FIXED10, refined
from mvo_amd import scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "monocular-visual-odometry_amd", "host", "adapter_check")

CASES = [
    dict(N=3, M=2, L=60, seed=7, span="full"),                      # the reference's defaults
    dict(N=8, M=3, L=50, seed=8, span=("anchored", 2, 8)),          # ragged tracks
    dict(N=12, M=5, L=80, seed=9, span=("sliding", 6)),             # tracks that start late
]


def _window(c):
    return scene.make_window(c["N"], c["M"], c["L"], c["seed"], span=c["span"], keep_tracks=True)


def _need_exe():
    if not os.path.exists(EXE):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(os.path.dirname(EXE)), "host/adapter_check"])


@pytest.mark.parametrize("c", CASES, ids=lambda c: f"n{c['N']}m{c['M']}")
def test_adapter_gathering_matches_generator_csr(tmp_path, c):
    _need_exe()
    w = _window(c)
    tracks, out = str(tmp_path / "tracks.bin"), str(tmp_path / "csr.bin")
    scene.write_track_file(w, tracks)
    subprocess.check_call([EXE, "gather", tracks, out])
    raw = open(out, "rb").read()
    n_pairs, n_blocks = struct.unpack_from("<2Q", raw, 0)
    assert n_pairs == w.n_pairs and n_blocks == w.n_residual_blocks
    offs = np.frombuffer(raw, dtype="<u8", count=n_pairs + 1, offset=16)
    base = 16 + 8 * (n_pairs + 1)
    m0 = np.frombuffer(raw, dtype="<f8", count=3 * n_blocks, offset=base).reshape(-1, 3)
    m1 = np.frombuffer(raw, dtype="<f8", count=3 * n_blocks, offset=base + 24 * n_blocks).reshape(-1, 3)
    assert np.array_equal(offs, w.pair_offsets)
    assert np.abs(m0 - w.m0).max() < 4e-16 and np.abs(m1 - w.m1).max() < 4e-16  # normFeatures of (x,y,1)


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES, ids=lambda c: f"n{c['N']}m{c['M']}")
@pytest.mark.parametrize("use_loss,fix", [(False, False), (True, False), (False, True)])
def test_adapter_refine_matches_oracle(tmp_path, oracle, c, use_loss, fix):
    _need_exe()
    w = _window(c)
    tracks, out = str(tmp_path / "tracks.bin"), str(tmp_path / "out.bin")
    scene.write_track_file(w, tracks, use_loss_function=use_loss, fix_length=fix, **FIXED10)
    subprocess.check_call([EXE, "refine", tracks, out])
    raw = open(out, "rb").read()
    got = np.frombuffer(raw, dtype="<f8", count=4 * w.N).reshape(w.N, 4)
    term, reason, iters = struct.unpack_from("<3i", raw, 32 * w.N)
    fo, so, _ = oracle.refine(w, oracle.make_options(use_loss_function=use_loss, fix_length=fix, **FIXED10))
    assert np.abs(got[: w.M] - refined(fo, w.M)).max() < 1e-6
    assert np.array_equal(got[w.M:], w.frames[w.M:, 9:13])  # frames >= M untouched (IR.cpp:240)
    # with tolerances 0 a stalled solve stops when a cost change is exactly 0.0; that iteration is
    # rounding dependent, so only runs that hit the cap are compared on the iteration count
    if so.reason == 1 and reason == 1:
        assert (term, iters) == (so.termination, so.iterations)


@pytest.mark.gpu
@pytest.mark.parametrize("use_loss", [False, True])
def test_staged_and_pageable_adapter_paths_agree_bit_for_bit(tmp_path, use_loss):
    """refine() gathers into the handle's pinned staging block (mvo_hip_staging / mvo_hip_refine_staged) by
    default; with staged = false it fills std::vectors and calls mvo_hip_refine.  Same blocks, same order,
    same kernels: the refined baselines and the summary must be byte-identical.  The second window is large
    enough that the staging block has to grow in the middle of the gather loop."""
    _need_exe()
    for c in (CASES[1], dict(N=30, M=5, L=1500, seed=1002, span=("sliding", 20))):
        w = _window(c)
        tracks = str(tmp_path / "tracks.bin")
        scene.write_track_file(w, tracks, use_loss_function=use_loss, **FIXED10)
        outs = []
        for mode in ("refine", "refine_pageable"):
            out = str(tmp_path / f"out_{mode}.bin")
            subprocess.check_call([EXE, mode, tracks, out])
            outs.append(open(out, "rb").read())
        assert outs[0] == outs[1] and len(outs[0]) == 32 * w.N + 40


def _read_csr(raw, base):
    n_pairs, n_blocks = struct.unpack_from("<2Q", raw, base)
    offs = np.frombuffer(raw, dtype="<u8", count=n_pairs + 1, offset=base + 16)
    b = base + 16 + 8 * (n_pairs + 1)
    m0 = np.frombuffer(raw, dtype="<f8", count=3 * n_blocks, offset=b).reshape(-1, 3)
    m1 = np.frombuffer(raw, dtype="<f8", count=3 * n_blocks, offset=b + 24 * n_blocks).reshape(-1, 3)
    return offs, m0, m1


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES, ids=lambda c: f"n{c['N']}m{c['M']}")
@pytest.mark.parametrize("use_loss", [False, True])
def test_track_mirror_equals_host_gathering_adapter(tmp_path, c, use_loss):
    """TrackMirror (keyframes mirrored into include/mvo_hip_tracks.h's table, gathering loop on the
    device) against the adapter that runs the reference's own loops on the host (`gather` / `refine`
    modes: Frame::getCorrespondingFeatures + normFeatures as C++): the CSR buffers and the refined
    baselines must be bit-identical."""
    _need_exe()
    w = _window(c)
    tracks = str(tmp_path / "tracks.bin")
    scene.write_track_file(w, tracks, use_loss_function=use_loss, **FIXED10)
    outs = {}
    # without a loss mvo_hip_tracks_refine sums the pair moments straight from the tables (other summation
    # order: rounding-level agreement); with the fusion off it is the same kernels on the same blocks
    env_nofuse = dict(os.environ, MVO_HIP_TRACKS_NO_FUSE="1")
    for mode in ("gather", "refine", "tracks"):
        outs[mode] = str(tmp_path / f"out_{mode}.bin")
        subprocess.check_call([EXE, mode, tracks, outs[mode]], env=env_nofuse if mode == "tracks" else None)
    host_csr = _read_csr(open(outs["gather"], "rb").read(), 0)
    host = open(outs["refine"], "rb").read()
    dev = open(outs["tracks"], "rb").read()
    n_res = 32 * w.N + 40  # N x (vec, scale) + mvo_hip_summary
    assert dev[:n_res] == host[:n_res]
    if not use_loss:
        fused_out = str(tmp_path / "out_tracks_fused.bin")
        subprocess.check_call([EXE, "tracks", tracks, fused_out])
        fused = open(fused_out, "rb").read()
        a = np.frombuffer(fused, dtype="<f8", count=4 * w.N)
        b = np.frombuffer(host, dtype="<f8", count=4 * w.N)
        assert np.abs(a - b).max() < 1e-8
    dev_csr = _read_csr(dev, n_res)
    for got, exp in zip(dev_csr, host_csr):
        assert np.array_equal(got, exp)
