"""The numpy restatement of the reference's host-side gather (scene.gather_tracks_numpy: chain walk of
sliding_window/Frame.hpp:277-291 per pair of IterativeRefinement.cpp:116-152, normFeatures
FeatureOperations.cpp:117-122) is the checker of the device-resident track tables
(tests/test_gpu_tracks.py).  It is pinned here against two independent constructions: the window
builder (which selects blocks from landmark visibility spans, no chains at all) and a literal,
object-per-feature transcription of the reference's loops."""
import numpy as np
import pytest

from mvo_amd import scene


@pytest.mark.parametrize("span", ["full", ("sliding", 4), ("anchored", 2, 8), ("anchored", 1, 2)], ids=str)
@pytest.mark.parametrize("unnormalised", [False, True])
def test_numpy_gather_equals_window_builder(span, unnormalised):
    w = scene.make_window(8, 3, 200, 5, span=span, keep_tracks=True)
    frames = scene.track_frames(w, unnormalised=unnormalised)
    offs, m0, m1 = scene.gather_tracks_numpy(frames, w.M)
    assert np.array_equal(offs, w.pair_offsets)
    if len(m0):
        assert np.abs(m0 - w.m0).max() <= 4e-16 and np.abs(m1 - w.m1).max() <= 4e-16


class _Feature:  # sliding_window/Feature.hpp:6-22
    def __init__(self, ray, pre):
        self.ray, self.pre, self.counter = ray, int(pre), 0


class _Frame:
    def __init__(self, rays, pre, pre_frame):
        self.features = [_Feature(r, p) for r, p in zip(rays, pre)]
        self.pre_frame = pre_frame
        if pre_frame is not None:  # Frame::calculateFeaturePreCounter, Frame.cpp:132-148
            for f in self.features:
                if f.pre >= 0:
                    f.counter = pre_frame.features[f.pre].counter + 1
                else:
                    f.counter, f.pre = 0, -1

    def previous(self, past):  # Frame::getPreviousFrame
        f = self
        for _ in range(past):
            f = f.pre_frame
        return f

    def corresponding(self, past):  # Frame.hpp:277-291
        old, new = [], []
        for f in self.features:
            if f.counter >= past:
                new.append(f.ray)
                o, fr = f, self
                for _ in range(past):
                    fr = fr.pre_frame
                    o = fr.features[o.pre]
                old.append(o.ray)
        return old, new


def _norm(v):  # FeatureOperations.cpp:117-122
    scale = 1.0 / np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
    return v * scale


def test_numpy_gather_equals_literal_transcription():
    w = scene.make_window(7, 3, 60, 21, span=("sliding", 5), keep_tracks=True)
    frames = scene.track_frames(w)
    now = None
    for rays, pre in frames:  # oldest first
        now = _Frame(rays, pre, now)
    offs, m0, m1 = [0], [], []
    for frame0 in range(w.M):  # IterativeRefinement.cpp:116,134
        for frame1 in range(frame0, w.N - 1):
            old, new = now.previous(frame0).corresponding(frame1 - frame0 + 1)  # :152
            m0 += [_norm(v) for v in new]
            m1 += [_norm(v) for v in old]
            offs.append(len(m0))
    go, g0, g1 = scene.gather_tracks_numpy(frames, w.M)
    assert np.array_equal(go, np.array(offs, dtype=np.uint64)) and offs[-1] > 0
    assert np.array_equal(g0, np.array(m0)) and np.array_equal(g1, np.array(m1))
