// IterativeRefinementHip.hpp -- the reference-side adapter: same signature as
// IterativeRefinement::refine (mvo/src/algorithms/IterativeRefinement.hpp:181-192), body replaced
// by one call into libmvo_hip.  C++11, warning-clean under -Wall -Werror -pedantic -Wextra
// (mvo/CMakeLists.txt:4,71).  Header-only; depends on include/mvo_hip.h and include/mvo_hip_tracks.h
// only.  Second half: TrackMirror, which keeps the Frame/Feature tables on the GPU so that the
// gathering loop runs there as well (include/mvo_hip_tracks.h).
//
// What stays on this side of the boundary (exactly the reference's own statements):
//   * packing refinementData -> plain records            (IterativeRefinement.cpp:36-43)
//   * the pair loops and the observation gathering       (IterativeRefinement.cpp:116-165):
//       nowFrame.getPreviousFrame(a).getCorrespondingFeatures(b - a + 1, feature1, feature0)
//       FeatureOperations::normFeatures(feature0 / feature1)   (FeatureOperations.cpp:117-122)
//     -- but the rays go into one CSR buffer instead of a std::list of std::vectors
//   * the write-back into refinementData                 (IterativeRefinement.cpp:240-246)
// What moves into the library: Ceres problem construction, solve, loss, parameterisation
//   (IterativeRefinement.cpp:63-109, 169-178, 231-237).
//
// Template parameters keep the header free of OpenCV/ROS: RefinementFrameT needs
//   R(i,j), vec[i], scale   (cv::Matx33d / cv::Vec3d / double in the reference);
// FrameT needs getPreviousFrame(unsigned), getCorrespondingFeatures(unsigned, std::vector<Ray>&,
//   std::vector<Ray>&) and getParameters().fixLength  (Frame.hpp:258-302, Frame.hpp:50).
#ifndef MVO_ITERATIVE_REFINEMENT_HIP_HPP
#define MVO_ITERATIVE_REFINEMENT_HIP_HPP

#include <cassert>
#include <cmath>
#include <cstddef>
#include <vector>

#include "mvo_hip.h"
#include "mvo_hip_tracks.h"

namespace mvo_hip_adapter {

struct GatheredObservations {
  std::vector<uint64_t> pair_offsets;  // |P| + 1
  std::vector<double> m0, m1;          // 3 doubles per residual block, unit rays
};

// FeatureOperations::normFeatures (FeatureOperations.cpp:117-122) fused with the copy into CSR
template <class RayT>
inline void appendNormalised(const std::vector<RayT> &rays, std::vector<double> &dst) {
  for (typename std::vector<RayT>::const_iterator f = rays.begin(); f != rays.end(); ++f) {
    const double x = (*f)[0], y = (*f)[1], z = (*f)[2];
    const double scale = 1.0 / std::sqrt(x * x + y * y + z * z);
    dst.push_back(x * scale);
    dst.push_back(y * scale);
    dst.push_back(z * scale);
  }
}

// IterativeRefinement.cpp:116-165 without the Ceres calls: same loops, same call into Frame.
template <class RayT, class FrameT>
inline void gather(const FrameT &nowFrame, unsigned int numberToRefine, unsigned int numberToNote,
                   GatheredObservations &out) {
  out.pair_offsets.clear();
  out.m0.clear();
  out.m1.clear();
  out.pair_offsets.push_back(0);
  for (unsigned int frame0 = 0; frame0 < numberToRefine; frame0++) {
    for (unsigned int frame1 = frame0; frame1 < (numberToNote - 1); frame1++) {
      std::vector<RayT> feature1, feature0;
      nowFrame.getPreviousFrame(frame0).getCorrespondingFeatures((frame1 - frame0) + 1, feature1, feature0);
      assert(feature0.size() == feature1.size());  // IterativeRefinement.cpp:379
      appendNormalised(feature0, out.m0);
      appendNormalised(feature1, out.m1);
      out.pair_offsets.push_back(static_cast<uint64_t>(out.m0.size() / 3));
    }
  }
}

// The same loops writing straight into the handle's PINNED staging block (mvo_hip_staging): the rays cross
// PCIe from where this loop put them -- no pageable copy by the HIP runtime, no second copy by the library.
// The block count is not known up front: room is asked for again (the block keeps what it holds) whenever
// a pair would not fit.
template <class RayT, class FrameT>
inline int gatherStaged(mvo_hip_handle handle, const FrameT &nowFrame, unsigned int numberToRefine,
                        unsigned int numberToNote) {
  const uint32_t nPairs = mvo_hip_num_pairs(numberToRefine, numberToNote);
  uint64_t *offsets = 0, capacity = 0, n = 0;
  double *m0 = 0, *m1 = 0;
  int rc = mvo_hip_staging(handle, nPairs, 0, &offsets, &m0, &m1, &capacity);
  if (rc != MVO_HIP_OK) return rc;
  uint32_t p = 0;
  offsets[0] = 0;
  for (unsigned int frame0 = 0; frame0 < numberToRefine; frame0++) {
    for (unsigned int frame1 = frame0; frame1 < (numberToNote - 1); frame1++) {
      std::vector<RayT> feature1, feature0;
      nowFrame.getPreviousFrame(frame0).getCorrespondingFeatures((frame1 - frame0) + 1, feature1, feature0);
      assert(feature0.size() == feature1.size());  // IterativeRefinement.cpp:379
      if (n + feature0.size() > capacity) {
        rc = mvo_hip_staging(handle, nPairs, 2 * (n + feature0.size()), &offsets, &m0, &m1, &capacity);
        if (rc != MVO_HIP_OK) return rc;
      }
      for (std::size_t i = 0; i < feature0.size(); i++, n++) {  // FeatureOperations::normFeatures, both sides
        const double x0 = feature0[i][0], y0 = feature0[i][1], z0 = feature0[i][2];
        const double s0 = 1.0 / std::sqrt(x0 * x0 + y0 * y0 + z0 * z0);
        m0[3 * n] = x0 * s0;
        m0[3 * n + 1] = y0 * s0;
        m0[3 * n + 2] = z0 * s0;
        const double x1 = feature1[i][0], y1 = feature1[i][1], z1 = feature1[i][2];
        const double s1 = 1.0 / std::sqrt(x1 * x1 + y1 * y1 + z1 * z1);
        m1[3 * n] = x1 * s1;
        m1[3 * n + 1] = y1 * s1;
        m1[3 * n + 2] = z1 * s1;
      }
      offsets[++p] = n;
    }
  }
  return MVO_HIP_OK;
}

// Drop-in body of IterativeRefinement::refine.  Returns the library status (the reference
// returns void and only logs on Ceres FAILURE; results are written back regardless,
// IterativeRefinement.cpp:235-246).  staged = true (default): the gather loop writes into the handle's
// pinned staging block and mvo_hip_refine_staged uploads from there; false: std::vectors + mvo_hip_refine
// (pageable memory: the HIP runtime copies the rays once more before they cross PCIe).
template <class RayT, class RefinementFrameT, class FrameT>
inline int refine(mvo_hip_handle handle, std::vector<RefinementFrameT> &refinementData,
                  unsigned int numberToRefine, unsigned int numberToNote, int maxNumthreads,
                  int maxNumIterations, double functionTolerance, double gradientTolerance,
                  double parameterTolerance, bool useLossFunction, double lowestLength, double highestLength,
                  const FrameT &nowFrame, mvo_hip_summary *summary = 0, bool staged = true) {
  assert(refinementData.size() == numberToNote);  // IterativeRefinement.cpp:28-30
  assert(numberToNote > numberToRefine);
  assert(numberToRefine > 0);

  std::vector<mvo_hip_frame> frames(refinementData.size());
  for (std::size_t i = 0; i < refinementData.size(); i++) {  // IterativeRefinement.cpp:36-43
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) frames[i].R[3 * r + c] = refinementData[i].R(r, c);
    for (int c = 0; c < 3; c++) frames[i].vec[c] = refinementData[i].vec[c];
    frames[i].scale = refinementData[i].scale;
  }

  mvo_hip_options opt;
  opt.max_num_iterations = maxNumIterations;
  opt.num_threads = maxNumthreads;
  opt.function_tolerance = functionTolerance;
  opt.gradient_tolerance = gradientTolerance;
  opt.parameter_tolerance = parameterTolerance;
  opt.use_loss_function = useLossFunction ? 1 : 0;
  opt.fix_length = nowFrame.getParameters().fixLength ? 1 : 0;  // IterativeRefinement.cpp:105
  opt.lowest_length = lowestLength;
  opt.highest_length = highestLength;
  opt.assembly_mode = MVO_HIP_ASSEMBLY_AUTO;
  opt.reserved = 0;

  int status;
  if (staged) {
    status = gatherStaged<RayT>(handle, nowFrame, numberToRefine, numberToNote);
    if (status == MVO_HIP_OK)
      status = mvo_hip_refine_staged(handle, frames.data(), numberToRefine, numberToNote, &opt, summary);
  } else {
    GatheredObservations g;
    gather<RayT>(nowFrame, numberToRefine, numberToNote, g);
    mvo_hip_observations obs;
    obs.n_pairs = static_cast<uint32_t>(g.pair_offsets.size() - 1);
    obs.reserved = 0;
    obs.pair_offsets = g.pair_offsets.data();
    obs.m0 = g.m0.data();
    obs.m1 = g.m1.data();
    status = mvo_hip_refine(handle, frames.data(), numberToRefine, numberToNote, &opt, &obs, summary);
  }
  if (status != MVO_HIP_OK) return status;

  for (unsigned int i = 0; i < numberToRefine; i++) {  // IterativeRefinement.cpp:240-246
    refinementData[i].scale = frames[i].scale;
    for (int c = 0; c < 3; c++) refinementData[i].vec[c] = frames[i].vec[c];
  }
  return MVO_HIP_OK;
}

// The Refiner's mirror of the sliding window on the GPU.  Every keyframe reaches Refiner::stage once
// (pipeline/Refiner.cpp:27), after the Merger fixed its connections (Merger.cpp:115-124) and the
// BaselineEstimator disbanded the outliers (BaselineEstimator.cpp:32): keyframe() pushes it then.
// Frame::mergeFrame (Frame.cpp:103-119) may append new features to a keyframe afterwards; they are
// picked up when the next keyframe arrives, before it is pushed.
// FrameT additionally needs
//   void getFeatureTable(std::size_t from, std::vector<RayT> &rays, std::vector<int> &preFeature) const
// returning _positionProjected and _preFeature of the features [from, size()) under the frame's lock.
template <class RayT, class FrameT>
class TrackMirror {
 public:
  TrackMirror() : _solver(0), _tracks(0), _pushedOfNewest(0) {}
  ~TrackMirror() {
    if (_tracks) mvo_hip_tracks_destroy(_tracks);
  }
  int create(mvo_hip_handle solver, unsigned int ringFrames, unsigned int maxFeatures) {
    _solver = solver;
    return mvo_hip_tracks_create(solver, ringFrames, maxFeatures, &_tracks);
  }
  mvo_hip_tracks_handle handle() const { return _tracks; }

  // newest keyframe; `previous` = the keyframe handed over by the call before (0 for the first)
  int keyframe(const FrameT &frame, const FrameT *previous) {
    std::vector<RayT> rays;
    std::vector<int> pre;
    if (previous) {
      previous->getFeatureTable(_pushedOfNewest, rays, pre);
      const int rc = mvo_hip_tracks_append(_tracks, static_cast<uint32_t>(rays.size()), flat(rays));
      if (rc != MVO_HIP_OK) return rc;
    }
    frame.getFeatureTable(0, rays, pre);
    std::vector<int32_t> pre32(pre.begin(), pre.end());
    _pushedOfNewest = rays.size();
    return mvo_hip_tracks_push(_tracks, static_cast<uint32_t>(rays.size()), flat(rays), pre32.data());
  }

  // IterativeRefinement::refine with the gathering loop on the device; same arguments, the window is
  // the newest numberToNote keyframes handed to keyframe()
  template <class RefinementFrameT>
  int refine(std::vector<RefinementFrameT> &refinementData, unsigned int numberToRefine, unsigned int numberToNote,
             int maxNumthreads, int maxNumIterations, double functionTolerance, double gradientTolerance,
             double parameterTolerance, bool useLossFunction, double lowestLength, double highestLength,
             bool fixLength, mvo_hip_summary *summary = 0) {
    assert(refinementData.size() == numberToNote);  // IterativeRefinement.cpp:28-30
    std::vector<mvo_hip_frame> frames(refinementData.size());
    for (std::size_t i = 0; i < refinementData.size(); i++) {  // IterativeRefinement.cpp:36-43
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) frames[i].R[3 * r + c] = refinementData[i].R(r, c);
      for (int c = 0; c < 3; c++) frames[i].vec[c] = refinementData[i].vec[c];
      frames[i].scale = refinementData[i].scale;
    }
    mvo_hip_options opt;
    opt.max_num_iterations = maxNumIterations;
    opt.num_threads = maxNumthreads;
    opt.function_tolerance = functionTolerance;
    opt.gradient_tolerance = gradientTolerance;
    opt.parameter_tolerance = parameterTolerance;
    opt.use_loss_function = useLossFunction ? 1 : 0;
    opt.fix_length = fixLength ? 1 : 0;  // IterativeRefinement.cpp:105
    opt.lowest_length = lowestLength;
    opt.highest_length = highestLength;
    opt.assembly_mode = MVO_HIP_ASSEMBLY_AUTO;
    opt.reserved = 0;
    const int status = mvo_hip_tracks_refine(_tracks, frames.data(), numberToRefine, numberToNote, &opt, summary);
    if (status != MVO_HIP_OK) return status;
    for (unsigned int i = 0; i < numberToRefine; i++) {  // IterativeRefinement.cpp:240-246
      refinementData[i].scale = frames[i].scale;
      for (int c = 0; c < 3; c++) refinementData[i].vec[c] = frames[i].vec[c];
    }
    return MVO_HIP_OK;
  }

 private:
  TrackMirror(const TrackMirror &);
  TrackMirror &operator=(const TrackMirror &);
  static const double *flat(const std::vector<RayT> &rays) {
    // cv::Vec3d is three contiguous doubles (cv::Matx<double, 3, 1>::val); so is the stand-in
    return rays.empty() ? 0 : &rays[0][0];
  }
  mvo_hip_handle _solver;
  mvo_hip_tracks_handle _tracks;
  std::size_t _pushedOfNewest;
};

}  // namespace mvo_hip_adapter

#endif  // MVO_ITERATIVE_REFINEMENT_HIP_HPP
