// adapter_check.cpp -- exercises host/IterativeRefinementHip.hpp with stand-ins for the reference's
// Frame / Feature data model (sliding_window/Frame.hpp, Feature.hpp), so that the adapter and its
// gathering order can be tested without OpenCV/ROS.
//
//   adapter_check gather <tracks.bin> <csr.bin>     CPU only: dumps the CSR buffers the adapter builds
//   adapter_check refine <tracks.bin> <out.bin>     GPU: runs refine() through libmvo_hip (gather into the handle's
//                                                   pinned staging block + mvo_hip_refine_staged)
//   adapter_check refine_pageable <..> <out.bin>    GPU: the same through std::vectors + mvo_hip_refine
//   adapter_check tracks <tracks.bin> <out.bin>     GPU: keyframes mirrored into the device-resident
//                                                   track table (TrackMirror), refine from there; the
//                                                   device-gathered CSR buffers follow the results
//
// tracks.bin (little endian): int32 N, M, maxNumIterations, useLoss, fixLength; double ftol, gtol,
//   ptol, lowest, highest; per frame: double R[9], vec[3], scale; int32 nfeat; per feature:
//   double ray[3] (x,y,1 as projectPixelTo3dRay returns), int32 preFeature, int32 preFeatureCounter.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "IterativeRefinementHip.hpp"

namespace {

struct Vec3 {
  double v[3];
  double &operator[](int i) { return v[i]; }
  const double &operator[](int i) const { return v[i]; }
};
struct Mat33 {
  double m[9];
  double operator()(int r, int c) const { return m[3 * r + c]; }
};
struct RefinementFrame {  // IterativeRefinement.hpp:171-175
  Mat33 R;
  Vec3 vec;
  double scale;
};
struct Feature {  // sliding_window/Feature.hpp
  Vec3 ray;
  int preFeature;
  unsigned int preFeatureCounter;
};
struct Parameters {
  bool fixLength;
};

// The subset of sliding_window/Frame.hpp that refine() touches, with the same semantics.
class Frame {
 public:
  std::vector<Feature> features;
  const Frame *preFrame;
  Parameters params;
  Frame() : preFrame(0) { params.fixLength = false; }
  const Parameters &getParameters() const { return params; }
  const Frame &getPreviousFrame(unsigned int past) const {
    const Frame *f = this;
    for (unsigned int i = 0; i < past; i++) f = f->preFrame;
    return *f;
  }
  // what TrackMirror reads: _positionProjected and _preFeature of the features [from, size())
  void getFeatureTable(std::size_t from, std::vector<Vec3> &rays, std::vector<int> &pre) const {
    rays.clear();
    pre.clear();
    for (std::size_t i = from; i < features.size(); i++) {
      rays.push_back(features[i].ray);
      pre.push_back(features[i].preFeature);
    }
  }
  // Frame.hpp:258-302: features of THIS frame whose track reaches `past` frames back
  void getCorrespondingFeatures(unsigned int past, std::vector<Vec3> &oldFeatures,
                                std::vector<Vec3> &newFeatures) const {
    for (std::size_t i = 0; i < features.size(); i++) {
      if (features[i].preFeatureCounter >= past) {
        newFeatures.push_back(features[i].ray);
        const Feature *old = &features[i];
        const Frame *frame = this;
        for (unsigned int k = 0; k < past; k++) {
          frame = frame->preFrame;
          old = &frame->features[static_cast<std::size_t>(old->preFeature)];
        }
        oldFeatures.push_back(old->ray);
      }
    }
  }
};

bool rd(FILE *f, void *p, std::size_t n) { return std::fread(p, 1, n, f) == n; }

}  // namespace

int main(int argc, char **argv) {
  if (argc != 4) {
    std::fprintf(stderr, "usage: adapter_check gather|refine|refine_pageable|tracks|latency <tracks.bin> <out.bin>\n");
    return 2;
  }
  const std::string mode = argv[1];
  FILE *f = std::fopen(argv[2], "rb");
  if (!f) return 3;
  int32_t hdr[5];
  double tol[5];
  if (!rd(f, hdr, sizeof hdr) || !rd(f, tol, sizeof tol)) return 4;
  const unsigned int N = static_cast<unsigned int>(hdr[0]), M = static_cast<unsigned int>(hdr[1]);
  std::vector<Frame> frames(N);
  std::vector<RefinementFrame> data(N);
  for (unsigned int i = 0; i < N; i++) {
    int32_t nfeat;
    if (!rd(f, data[i].R.m, 72) || !rd(f, data[i].vec.v, 24) || !rd(f, &data[i].scale, 8) || !rd(f, &nfeat, 4))
      return 4;
    frames[i].features.resize(static_cast<std::size_t>(nfeat));
    frames[i].params.fixLength = hdr[4] != 0;
    frames[i].preFrame = (i + 1 < N) ? &frames[i + 1] : 0;
    for (int32_t k = 0; k < nfeat; k++) {
      Feature &ft = frames[i].features[static_cast<std::size_t>(k)];
      int32_t pre, cnt;
      if (!rd(f, ft.ray.v, 24) || !rd(f, &pre, 4) || !rd(f, &cnt, 4)) return 4;
      ft.preFeature = pre;
      ft.preFeatureCounter = static_cast<unsigned int>(cnt);
    }
  }
  std::fclose(f);

  FILE *o = std::fopen(argv[3], "wb");
  if (!o) return 5;
  if (mode == "gather") {
    mvo_hip_adapter::GatheredObservations g;
    mvo_hip_adapter::gather<Vec3>(frames[0], M, N, g);
    const uint64_t np = g.pair_offsets.size() - 1, nb = g.m0.size() / 3;
    std::fwrite(&np, 8, 1, o);
    std::fwrite(&nb, 8, 1, o);
    std::fwrite(g.pair_offsets.data(), 8, g.pair_offsets.size(), o);
    std::fwrite(g.m0.data(), 8, g.m0.size(), o);
    std::fwrite(g.m1.data(), 8, g.m1.size(), o);
  } else if (mode == "latency") {
    // what the reference's Refiner::stage would wait for per keyframe: the adapter's refine() (host chain
    // walk + mvo_hip_refine) and mvo_hip_refine alone on the gathered blocks, from C++, median of 300 calls
    mvo_hip_handle h = 0;
    int rc = mvo_hip_create(0, 1, N, 0, &h);
    if (rc != MVO_HIP_OK) return 10 + rc;
    const std::vector<RefinementFrame> data0 = data;
    mvo_hip_summary s;
    std::vector<double> t_all, t_call, t_pageable;
    for (int staged = 1; staged >= 0; staged--)
      for (int it = 0; it < 320 && rc == MVO_HIP_OK; it++) {
        data = data0;
        const std::chrono::steady_clock::time_point a = std::chrono::steady_clock::now();
        rc = mvo_hip_adapter::refine<Vec3>(h, data, M, N, 4, hdr[2], tol[0], tol[1], tol[2], hdr[3] != 0, tol[3], tol[4],
                                           frames[0], &s, staged != 0);
        const std::chrono::steady_clock::time_point b = std::chrono::steady_clock::now();
        if (it >= 20) (staged ? t_all : t_pageable).push_back(std::chrono::duration<double, std::micro>(b - a).count());
      }
    mvo_hip_adapter::GatheredObservations g;
    mvo_hip_adapter::gather<Vec3>(frames[0], M, N, g);
    mvo_hip_observations obs;
    obs.n_pairs = static_cast<uint32_t>(g.pair_offsets.size() - 1);
    obs.reserved = 0;
    obs.pair_offsets = g.pair_offsets.data();
    obs.m0 = g.m0.data();
    obs.m1 = g.m1.data();
    mvo_hip_options opt;
    opt.max_num_iterations = hdr[2];
    opt.num_threads = 4;
    opt.function_tolerance = tol[0];
    opt.gradient_tolerance = tol[1];
    opt.parameter_tolerance = tol[2];
    opt.use_loss_function = hdr[3] != 0;
    opt.fix_length = hdr[4] != 0;
    opt.lowest_length = tol[3];
    opt.highest_length = tol[4];
    opt.assembly_mode = MVO_HIP_ASSEMBLY_AUTO;
    opt.reserved = 0;
    std::vector<mvo_hip_frame> fr0(N), fr(N);
    for (unsigned int i = 0; i < N; i++) {
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) fr0[i].R[3 * r + c] = data0[i].R(r, c);
      for (int c = 0; c < 3; c++) fr0[i].vec[c] = data0[i].vec[c];
      fr0[i].scale = data0[i].scale;
    }
    for (int it = 0; it < 320 && rc == MVO_HIP_OK; it++) {
      fr = fr0;
      const std::chrono::steady_clock::time_point a = std::chrono::steady_clock::now();
      rc = mvo_hip_refine(h, fr.data(), M, N, &opt, &obs, &s);
      const std::chrono::steady_clock::time_point b = std::chrono::steady_clock::now();
      if (it >= 20) t_call.push_back(std::chrono::duration<double, std::micro>(b - a).count());
    }
    // ... and the same window through the device-resident track tables (TrackMirror::refine)
    std::vector<double> t_mirror;
    {
      std::size_t widest = 1;
      for (unsigned int i = 0; i < N; i++) widest = frames[i].features.size() > widest ? frames[i].features.size() : widest;
      mvo_hip_adapter::TrackMirror<Vec3, Frame> mirror;
      if (rc == MVO_HIP_OK) rc = mirror.create(h, N + 1, static_cast<unsigned int>(widest));
      for (unsigned int k = N; rc == MVO_HIP_OK && k-- > 0;) rc = mirror.keyframe(frames[k], k + 1 < N ? &frames[k + 1] : 0);
      for (int it = 0; it < 320 && rc == MVO_HIP_OK; it++) {
        data = data0;
        const std::chrono::steady_clock::time_point a = std::chrono::steady_clock::now();
        rc = mirror.refine(data, M, N, 4, hdr[2], tol[0], tol[1], tol[2], hdr[3] != 0, tol[3], tol[4], hdr[4] != 0, &s);
        const std::chrono::steady_clock::time_point b = std::chrono::steady_clock::now();
        if (it >= 20) t_mirror.push_back(std::chrono::duration<double, std::micro>(b - a).count());
      }
    }
    mvo_hip_destroy(h);
    if (rc != MVO_HIP_OK) {
      std::fprintf(stderr, "latency: %s\n", mvo_hip_status_string(rc));
      return 10 + rc;
    }
    std::sort(t_all.begin(), t_all.end());
    std::sort(t_call.begin(), t_call.end());
    std::sort(t_mirror.begin(), t_mirror.end());
    std::sort(t_pageable.begin(), t_pageable.end());
    std::fprintf(o, "adapter refine (gather into pinned staging + mvo_hip_refine_staged): median %.1f us, p90 %.1f us; the same through std::vectors + mvo_hip_refine: median %.1f us; mvo_hip_refine alone (blocks already gathered, pageable): median %.1f us, p90 %.1f us; TrackMirror::refine (device-resident tracks): median %.1f us, p90 %.1f us; %d iterations, %u residual blocks\n",
                 t_all[t_all.size() / 2], t_all[t_all.size() * 9 / 10], t_pageable[t_pageable.size() / 2], t_call[t_call.size() / 2],
                 t_call[t_call.size() * 9 / 10], t_mirror[t_mirror.size() / 2], t_mirror[t_mirror.size() * 9 / 10],
                 s.iterations, static_cast<unsigned int>(g.m0.size() / 3));
  } else if (mode == "tracks") {
    mvo_hip_handle h = 0;
    int rc = mvo_hip_create(0, 1, N, 0, &h);
    if (rc != MVO_HIP_OK) {
      std::fprintf(stderr, "mvo_hip_create: %s\n", mvo_hip_status_string(rc));
      return 10 + rc;
    }
    std::size_t widest = 1;
    for (unsigned int i = 0; i < N; i++) widest = frames[i].features.size() > widest ? frames[i].features.size() : widest;
    mvo_hip_summary s;
    std::memset(&s, 0, sizeof s);
    std::vector<uint64_t> offs;
    std::vector<double> m0, m1;
    {
      mvo_hip_adapter::TrackMirror<Vec3, Frame> mirror;
      rc = mirror.create(h, N + 1, static_cast<unsigned int>(widest));
      for (unsigned int k = N; rc == MVO_HIP_OK && k-- > 0;)  // oldest keyframe first, as the pipeline delivers them
        rc = mirror.keyframe(frames[k], k + 1 < N ? &frames[k + 1] : 0);
      if (rc == MVO_HIP_OK)
        rc = mirror.refine(data, M, N, 4, hdr[2], tol[0], tol[1], tol[2], hdr[3] != 0, tol[3], tol[4], hdr[4] != 0, &s);
      if (rc == MVO_HIP_OK) {
        offs.resize(mvo_hip_num_pairs(M, N) + 1);
        rc = mvo_hip_tracks_gather(mirror.handle(), M, N, offs.data(), 0, 0, 0);
      }
      if (rc == MVO_HIP_OK) {
        m0.resize(3 * offs.back() + 3);
        m1.resize(3 * offs.back() + 3);
        rc = mvo_hip_tracks_gather(mirror.handle(), M, N, offs.data(), m0.data(), m1.data(), offs.back());
      }
    }
    mvo_hip_destroy(h);
    if (rc != MVO_HIP_OK) {
      std::fprintf(stderr, "tracks: %s\n", mvo_hip_status_string(rc));
      return 10 + rc;
    }
    for (unsigned int i = 0; i < N; i++) {
      std::fwrite(data[i].vec.v, 8, 3, o);
      std::fwrite(&data[i].scale, 8, 1, o);
    }
    std::fwrite(&s, sizeof s, 1, o);
    const uint64_t np = offs.size() - 1, nb = offs.back();
    std::fwrite(&np, 8, 1, o);
    std::fwrite(&nb, 8, 1, o);
    std::fwrite(offs.data(), 8, offs.size(), o);
    std::fwrite(m0.data(), 8, 3 * nb, o);
    std::fwrite(m1.data(), 8, 3 * nb, o);
  } else {
    mvo_hip_handle h = 0;
    int rc = mvo_hip_create(0, 1, N, 0, &h);
    if (rc != MVO_HIP_OK) {
      std::fprintf(stderr, "mvo_hip_create: %s\n", mvo_hip_status_string(rc));
      return 10 + rc;
    }
    mvo_hip_summary s;
    std::memset(&s, 0, sizeof s);
    rc = mvo_hip_adapter::refine<Vec3>(h, data, M, N, 4, hdr[2], tol[0], tol[1], tol[2], hdr[3] != 0, tol[3],
                                       tol[4], frames[0], &s, mode != "refine_pageable");
    mvo_hip_destroy(h);
    if (rc != MVO_HIP_OK) {
      std::fprintf(stderr, "refine: %s\n", mvo_hip_status_string(rc));
      return 10 + rc;
    }
    for (unsigned int i = 0; i < N; i++) {
      std::fwrite(data[i].vec.v, 8, 3, o);
      std::fwrite(&data[i].scale, 8, 1, o);
    }
    std::fwrite(&s, sizeof s, 1, o);
  }
  std::fclose(o);
  return 0;
}
