#!/usr/bin/env python3
"""Writes integration/mvo_use_hip.patch: the four reference-side edits of INTEGRATION.md as a unified
diff against the reference tree (run where /root/reference exists; the patch only ADDS lines, so the
only reference text it carries is two lines of context around each insertion).
  python scripts/make_integration_patch.py            # regenerate
  python scripts/make_integration_patch.py --check    # does the committed patch still apply?"""
import difflib
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("MVO_REFERENCE", "/root/reference")
PATCH = os.path.join(ROOT, "integration", "mvo_use_hip.patch")

CMAKE = '''
# --- libmvo_hip: refine on an AMD GPU (MI355X, gfx950) instead of Ceres -------------------------
option(MVO_USE_HIP "refine on an AMD GPU through libmvo_hip" OFF)
if(MVO_USE_HIP)
  find_path(MVO_HIP_INCLUDE_DIR mvo_hip.h)                   # <libmvo_hip repo>/include
  find_path(MVO_HIP_ADAPTER_DIR IterativeRefinementHip.hpp)  # <libmvo_hip repo>/monocular-visual-odometry_amd/host
  find_library(MVO_HIP_LIBRARY mvo_hip)                      # <libmvo_hip repo>/monocular-visual-odometry_amd
  target_include_directories(mvo_node PRIVATE ${MVO_HIP_INCLUDE_DIR} ${MVO_HIP_ADAPTER_DIR})
  target_link_libraries(mvo_node ${MVO_HIP_LIBRARY})
  target_compile_definitions(mvo_node PRIVATE MVO_USE_HIP)
endif()
'''

HPP_INCLUDE = '''#ifdef MVO_USE_HIP
#include <stdexcept>
#include "mvo_hip.h"
#endif
'''

HPP_BEFORE_CTOR = '''#ifdef MVO_USE_HIP
  // one handle for the life of the node: refine() is only ever called from _refinerThread (mvo.cpp:37)
  mvo_hip_handle _hip = nullptr;

  IterativeRefinement() {
    if (mvo_hip_create(0, 1, 100, 0, &_hip) != MVO_HIP_OK) throw std::runtime_error("libmvo_hip: no usable GPU");
  }

  ~IterativeRefinement() { mvo_hip_destroy(_hip); }

  IterativeRefinement(const IterativeRefinement &) = delete;

  IterativeRefinement &operator=(const IterativeRefinement &) = delete;
#else
'''

CPP_INCLUDE = '''#ifdef MVO_USE_HIP
#include "IterativeRefinementHip.hpp"
#endif
'''

CPP_BODY = '''
#ifdef MVO_USE_HIP
  {
    // libmvo_hip replaces the Ceres problem set-up, ceres::Solve and the write-back below; the adapter
    // keeps this function's own packing and gathering loops (getCorrespondingFeatures + normFeatures).
    mvo_hip_summary summary;
    const int status = mvo_hip_adapter::refine<cv::Vec3d>(
        _hip, refinementData, numberToRefine, numberToNote, maxNumthreads, maxNumIterations, functionTolerance,
        gradientTolerance, parameterTolerance, useLossFunction, lowestLength, highestLength, nowFrame, &summary);
    if (status != MVO_HIP_OK) {
      ROS_ERROR_STREAM("libmvo_hip: " << mvo_hip_status_string(status));
      return;
    }
    for (unsigned int i = 0; i < numberToRefine; i++) {  // the warnings reverseScale would have logged
      if (summary.clamped_scales & (1 << (2 * i))) ROS_WARN_STREAM("Lower Bound: frame " << i << " <= " << lowestLength);
      if (summary.clamped_scales & (1 << (2 * i + 1))) ROS_WARN_STREAM("Upper Bound: frame " << i << " >= " << highestLength);
    }
    if (summary.termination == MVO_HIP_FAILURE) {
      ROS_ERROR_STREAM("CERES ERROR! (libmvo_hip, reason " << summary.reason << ")");
    }
    for (unsigned int i = 0; i < numberToRefine; i++) {
      ROS_INFO_STREAM("[" << i << "] - " << refinementData[i].scale << " * " << refinementData[i].vec);
    }
    ROS_INFO_STREAM("libmvo_hip: " << summary.iterations << " iterations, cost " << summary.initial_cost << " -> "
                                   << summary.final_cost);
    return;
  }
#endif
'''


FRAME_HPP = '''#ifdef MVO_USE_HIP
  /**
   * What mvo_hip_adapter::TrackMirror reads of a keyframe (IterativeRefinementHip.hpp): the projected
   * positions and the predecessor indices of the features [from, size()), under the frame's lock.
   */
  void getFeatureTable(std::size_t from, std::vector<cv::Vec3d> &rays, std::vector<int> &preFeature) const {
    rays.clear();
    preFeature.clear();
    this->lock();
    for (std::size_t i = from; i < this->_features.size(); i++) {
      rays.push_back(this->_features[i]._positionProjected);
      preFeature.push_back(this->_features[i]._preFeature);
    }
    this->unlock();
  }
#endif

'''


def insert_after(lines, needle, text, occurrence=0):
    idx = [i for i, l in enumerate(lines) if needle in l][occurrence]
    return lines[: idx + 1] + text.splitlines(keepends=True) + lines[idx + 1:]


def insert_before(lines, needle, text, back=0):
    idx = [i for i, l in enumerate(lines) if needle in l][0] - back
    return lines[:idx] + text.splitlines(keepends=True) + lines[idx:]


def edited(rel, lines):
    if rel.endswith("CMakeLists.txt"):
        return insert_after(lines, "add_dependencies(mvo_node mvo_gencfg)", CMAKE)
    if rel.endswith("Frame.hpp"):
        # in front of the doc comment of the member getCorrespondingFeatures(past, ...)
        return insert_before(lines, " * Retrieves corresponding Features between this Frame and the n to the past Frame.", FRAME_HPP, back=1)
    if rel.endswith("IterativeRefinement.hpp"):
        lines = insert_after(lines, '#include "../operations/VisualisationUtils.hpp"', HPP_INCLUDE)
        lines = insert_before(lines, "IterativeRefinement() = default;", HPP_BEFORE_CTOR)
        return insert_after(lines, "~IterativeRefinement() = default;", "#endif\n")
    lines = insert_after(lines, "#include <list>", CPP_INCLUDE)
    return insert_after(lines, "assert(numberToRefine > 0);", CPP_BODY)


FILES = ["mvo/CMakeLists.txt", "mvo/src/algorithms/IterativeRefinement.hpp", "mvo/src/algorithms/IterativeRefinement.cpp",
         "mvo/src/sliding_window/Frame.hpp"]


def build():
    out = []
    for rel in FILES:
        with open(os.path.join(REF, rel), newline="") as f:
            a = f.readlines()
        if a and not a[-1].endswith("\n"):
            a[-1] += "\n"  # (the header has no final newline; the hunks below never reach the last line)
        b = edited(rel, list(a))
        out += list(difflib.unified_diff(a, b, "a/" + rel, "b/" + rel, n=2))
    return "".join(out)


def check():
    """Applies the committed patch to a scratch copy of the three files; returns git apply's status."""
    with tempfile.TemporaryDirectory() as tmp:
        for rel in FILES:
            os.makedirs(os.path.dirname(os.path.join(tmp, rel)), exist_ok=True)
            shutil.copy(os.path.join(REF, rel), os.path.join(tmp, rel))
        subprocess.check_call(["git", "init", "-q"], cwd=tmp)
        r = subprocess.run(["git", "apply", "--check", "--verbose", PATCH], cwd=tmp, capture_output=True, text=True)
        if r.returncode == 0:
            subprocess.check_call(["git", "apply", PATCH], cwd=tmp)
            body = open(os.path.join(tmp, FILES[2])).read()
            assert "mvo_hip_adapter::refine<cv::Vec3d>" in body and body.count("#ifdef MVO_USE_HIP") == 2
            assert "void getFeatureTable(std::size_t from" in open(os.path.join(tmp, FILES[3])).read()
        return r.returncode, r.stderr


if __name__ == "__main__":
    if "--check" in sys.argv:
        rc, err = check()
        print(err.strip() or "applies cleanly")
        sys.exit(rc)
    with open(PATCH, "w", newline="") as f:
        f.write(build())
    print("wrote", PATCH)
