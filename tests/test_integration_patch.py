"""integration/mvo_use_hip.patch: the reference-side edits of INTEGRATION.md as a unified diff.
Without the reference tree (the GPU box) only the patch's shape is checked; where /root/reference
exists the patch is applied to a scratch copy of the three files it touches (git apply)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCH = os.path.join(ROOT, "integration", "mvo_use_hip.patch")


def test_patch_only_adds_lines_to_the_four_files():
    lines = open(PATCH).read().splitlines()
    files = [l[6:] for l in lines if l.startswith("+++ b/")]
    assert files == ["mvo/CMakeLists.txt", "mvo/src/algorithms/IterativeRefinement.hpp",
                     "mvo/src/algorithms/IterativeRefinement.cpp", "mvo/src/sliding_window/Frame.hpp"]
    removed = [l for l in lines if l.startswith("-") and not l.startswith("--- ")]
    assert removed == []  # nothing of the reference is deleted: the Ceres path stays behind #ifdef / an early return
    added = "\n".join(l[1:] for l in lines if l.startswith("+") and not l.startswith("+++ "))
    for needle in ("option(MVO_USE_HIP", "find_library(MVO_HIP_LIBRARY mvo_hip)", "mvo_hip_create(0, 1, 100, 0, &_hip)",
                   "mvo_hip_adapter::refine<cv::Vec3d>", "summary.clamped_scales", "MVO_HIP_FAILURE",
                   "void getFeatureTable(std::size_t from, std::vector<cv::Vec3d> &rays, std::vector<int> &preFeature) const"):
        assert needle in added, needle


@pytest.mark.skipif(not os.path.isdir("/root/reference/mvo"), reason="the reference tree is not on this machine")
def test_patch_applies_to_the_reference():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "make_integration_patch.py"), "--check"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
