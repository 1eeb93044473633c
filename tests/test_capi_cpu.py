"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol that
include/mvo_hip.h declares, its records have the layouts the reference-side binding assumes,
and -- on a machine without a GPU -- it refuses to run instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from mvo_amd import capi, scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mvo_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mvo_hip_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mvo_hip.h but not exported"
    assert sorted(capi.EXPORTED_SYMBOLS) == names


def test_ba_header_symbols_are_exported_and_bound():
    from mvo_amd import ba_capi

    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mvo_hip_ba.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(mvo_hip_ba_[a-z_]+)\s*\(", text)))
    lib = ba_capi.load()
    assert len(names) == 17 and sorted(ba_capi.EXPORTED_SYMBOLS) == names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mvo_hip_ba.h but not exported"
    assert C.sizeof(ba_capi.Options) == 48 and C.sizeof(ba_capi.Summary) == 32
    assert C.sizeof(ba_capi.Iteration) == 56 and C.sizeof(ba_capi.Timing) == 64
    with pytest.raises(capi.MvoHipError) as e:
        ba_capi.BaSolver(0)
    assert e.value.status == capi.ERR_NO_DEVICE


def test_scaler_header_symbols_are_exported_and_bound():
    from mvo_amd import scaler_capi

    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mvo_hip_scaler.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(mvo_hip_scaler_[a-z_]+)\s*\(", text)))
    lib = scaler_capi.load()
    assert len(names) == 7 and sorted(scaler_capi.EXPORTED_SYMBOLS) == names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mvo_hip_scaler.h but not exported"
    assert C.sizeof(scaler_capi.Input) == 264 and C.sizeof(scaler_capi.Result) == 32
    with pytest.raises(capi.MvoHipError) as e:
        scaler_capi.Scaler(0)
    assert e.value.status == capi.ERR_NO_DEVICE


def test_epipolar_header_symbols_are_exported_and_bound():
    from mvo_amd import epipolar_capi

    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mvo_hip_epipolar.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(mvo_hip_epipolar_[a-z_]+)\s*\(", text)))
    lib = epipolar_capi.load()
    assert len(names) == 5 and sorted(epipolar_capi.EXPORTED_SYMBOLS) == names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mvo_hip_epipolar.h but not exported"
    assert C.sizeof(epipolar_capi.Result) == 56
    with pytest.raises(capi.MvoHipError) as e:
        epipolar_capi.Epipolar(0)
    assert e.value.status == capi.ERR_NO_DEVICE


def test_tracks_header_symbols_are_exported_and_bound():
    from mvo_amd import tracks_capi

    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "mvo_hip_tracks.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(mvo_hip_tracks_[a-z_]+)\s*\(", text)))
    lib = tracks_capi.load()
    assert len(names) == 11 and sorted(tracks_capi.EXPORTED_SYMBOLS) == names
    assert C.sizeof(tracks_capi.Timing) == 32
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mvo_hip_tracks.h but not exported"
    # a table lives on a solver's GPU: without one there is nothing to bind to (and no CPU path)
    h = C.c_void_p()
    assert lib.mvo_hip_tracks_create(None, 8, 100, C.byref(h)) == capi.ERR_INVALID_ARGUMENT and not h.value


def test_record_layouts():
    # mvo_hip_frame = cv::Matx33d R + cv::Vec3d vec + double scale (IterativeRefinement.hpp:171-175)
    assert scene.make_window(3, 2, 5, 1).frames.shape[1] * 8 == 104
    assert C.sizeof(capi.Options) == 64
    assert C.sizeof(capi.Observations) == 32
    assert C.sizeof(capi.Summary) == 40
    assert C.sizeof(capi.Iteration) == 56
    assert C.sizeof(capi.Timing) == 64


def test_header_is_plain_c_and_pedantic_cxx11(tmp_path):
    """The reference builds with -std=c++11 -Wall -Werror -pedantic -Wextra (mvo/CMakeLists.txt:4,71)."""
    import subprocess

    for compiler, std, name in (("g++", "-std=c++11", "t.cpp"), ("gcc", "-std=c99", "t.c")):
        src = tmp_path / name
        src.write_text('#include "mvo_hip.h"\n#include "mvo_hip_ba.h"\n#include "mvo_hip_scaler.h"\n#include "mvo_hip_epipolar.h"\n#include "mvo_hip_tracks.h"\n'
                       'int main(void){ mvo_hip_options o; mvo_hip_ba_options b; (void)o; (void)b; return 0; }\n')
        subprocess.check_call([compiler, std, "-Wall", "-Werror", "-pedantic", "-Wextra", "-fsyntax-only",
                               "-I", os.path.join(ROOT, "include"), str(src)])


def test_host_scalar_helpers_match_reference_formulas():
    # reverseScale / scaleTemplated (IterativeRefinement.cpp:298-309,348-368)
    for n in (0.3, 0.9, 1.7):
        t = capi.reverse_scale(n)
        assert abs(t + np.log((2.0 - n) / (n - 0.25))) < 1e-15
        assert abs(capi.scale(t) - n) < 1e-15
    assert capi.scale(-800.0) == 0.25
    eps = np.finfo(float).eps
    assert capi.reverse_scale(0.1) == capi.reverse_scale(0.25 + 0.1 * eps + eps)
    assert capi.reverse_scale(9.0) == capi.reverse_scale(2.0 - (9.0 * eps + eps))
    assert capi.num_pairs(5, 10) == 35 and capi.num_pairs(2, 3) == 3 and capi.num_pairs(3, 3) == 0
    assert b"gfx950" in capi.load().mvo_hip_version()


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a machine without a GPU")
def test_no_gpu_means_loud_failure_not_cpu_fallback():
    with pytest.raises(capi.MvoHipError) as e:
        capi.Solver(0)
    assert e.value.status == capi.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "monocular-visual-odometry_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")) or f == "Makefile":
                text = open(os.path.join(d, f), errors="ignore").read()
                assert "mvo_oracle" not in text and "oracle/" not in text, os.path.join(d, f)
