// mvo_hip.hip -- host side of libmvo_hip: the C ABI declared in include/mvo_hip.h.
//
// Replaces IterativeRefinement::refine (/root/reference/mvo/src/algorithms/
// IterativeRefinement.cpp:15-250, "IR.cpp") from the solver set-up to the write-back.
// There is no CPU fallback: without a HIP device every entry point fails.
#include "../../include/mvo_hip.h"

#include <hip/hip_runtime.h>

#include "mvo_alloc.h"
#include "mvo_rccl.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <unordered_map>
#include <new>
#include <vector>

#include "mvo_kernels.h"
#include "mvo_lm.h"
#include "mvo_lm4.h"

static_assert(sizeof(mvo_hip_frame) == MVO_FRAME_DOUBLES * sizeof(double), "frame record layout");
static_assert(sizeof(TraceRec) == sizeof(mvo_hip_iteration), "trace record layout");
static_assert(sizeof(LmState) % 8 == 0, "LmState is copied as doubles");
static_assert(MVO_HIP_MAX_REFINE == MVO_MAXM, "M limit");

// ---- csrc/mvo_alloc.h ------------------------------------------------------------------------
namespace {
constexpr size_t kGuardBytes = 256;
constexpr unsigned char kGuardByte = 0xA5;
struct AllocRegistry {
  std::mutex mu;
  std::unordered_map<void*, size_t> live;  // poison mode only: user bytes of every live allocation
  long violations = 0;
};
AllocRegistry& alloc_registry() {
  static AllocRegistry r;
  return r;
}
bool alloc_poison() {
  static const bool on = std::getenv("MVO_HIP_POISON_ALLOC") != nullptr;
  return on;
}
// true when the guard behind (p, bytes) is intact; the device must be idle
bool guard_intact(const void* p, size_t bytes) {
  unsigned char g[kGuardBytes];
  if (hipMemcpy(g, static_cast<const char*>(p) + bytes, kGuardBytes, hipMemcpyDeviceToHost) != hipSuccess) return false;
  for (size_t i = 0; i < kGuardBytes; ++i)
    if (g[i] != kGuardByte) {
      std::fprintf(stderr, "libmvo_hip: guard behind a %zu-byte device allocation overwritten at +%zu\n", bytes, i);
      return false;
    }
  return true;
}
}  // namespace

// resolves this device's address of the no-alloc entry point's scratch array (defined at the end of the file)
hipError_t mvo_noalloc_resolve(int device);

hipError_t mvo_device_alloc(void** p, size_t bytes) {
  if (!alloc_poison()) return hipMalloc(p, bytes);
  hipError_t e = hipMalloc(p, bytes + kGuardBytes);
  if (e != hipSuccess) return e;
  // hipMemset runs on the null stream and may return before it is done; the library's own streams
  // are non-blocking ones that do not wait for it: drain the device before handing the memory out
  e = hipMemset(*p, 0xFF, bytes);
  if (e == hipSuccess) e = hipMemset(static_cast<char*>(*p) + bytes, kGuardByte, kGuardBytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)hipFree(*p);
    *p = nullptr;
    return e;
  }
  AllocRegistry& r = alloc_registry();
  std::lock_guard<std::mutex> lock(r.mu);
  r.live[*p] = bytes;
  return hipSuccess;
}

hipError_t mvo_device_free(void* p) {
  if (alloc_poison() && p) {
    AllocRegistry& r = alloc_registry();
    size_t bytes = 0;
    bool known = false;
    {
      std::lock_guard<std::mutex> lock(r.mu);
      auto it = r.live.find(p);
      if (it != r.live.end()) {
        known = true;
        bytes = it->second;
        r.live.erase(it);
      }
    }
    if (known) {
      (void)hipDeviceSynchronize();
      if (!guard_intact(p, bytes)) {
        std::lock_guard<std::mutex> lock(r.mu);
        ++r.violations;
      }
    }
  }
  return hipFree(p);
}

namespace {

#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) {                             \
      h->last_hip_error = e_;                           \
      return e_ == hipErrorOutOfMemory ? MVO_HIP_ERR_OUT_OF_MEMORY : MVO_HIP_ERR_DEVICE; \
    }                                                   \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)mvo_device_free(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = mvo_device_alloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)mvo_device_free(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const { return static_cast<T*>(p); }
};

constexpr int kMaxTimedLaunches = 128;
constexpr int kMaxOverlapChunks = 8;
constexpr uint32_t kLm4FoldWavesPerCu = 512;  // fold wavefronts per CU in front of k_lm4 (many rounds, no ticket tails: mvo_hip_upload)
constexpr int kOverlapChunks = 4;  // a k_lm launch over W / 4 windows takes about as long as a quarter of the fold
// uploads with at most this many residual blocks stage their rays in the pinned arena too
constexpr uint64_t kArenaRayLimit = 1u << 16;
// windows with at most this many residual blocks are evaluated inside k_lm
constexpr uint64_t kInlineWindowLimit = 8192;
constexpr uint64_t kInlinePassBatchLimit = 2048;  // ... batches under the loss, up to one window per CU (mvo_hip_upload)
constexpr uint64_t kInlineFusedLimit = 3072;  // ... when the alternative is the fused fold + LM launch (mvo_hip_upload)

}  // namespace

struct mvo_hip_context {
  int device = 0;
  hipStream_t stream = nullptr;
  hipError_t last_hip_error = hipSuccess;
  // device buffers
  DevBuf d_arena, d_m0, d_m1, d_c, d_partial, d_pair_u, d_states, d_ctl, d_trace, d_res, d_jac, d_pairsum, d_pticket,
      d_proj,  // pass mode: T(6) q(3) cost(1) per pair, left by the passes' tails (project_pair8); twice: the
               // persistent robust solver (k_lm<256, 6>) alternates between the two halves
      d_grid,  // its barrier counter and give-up flag (mvo_grid.h)
      d_cls;   // partitioned pass mode: MVO_CLS_DOUBLES class totals per window (k_class_sums), what the ranks all-reduce
  // The tables of one upload (and the rays of a small one) are staged in ONE pinned block and
  // reach the device in ONE copy into d_arena; these point into it (rays: or into d_m0 / d_m1).
  struct {
    double* frames = nullptr;
    double* initx = nullptr;
    WinDev* wins = nullptr;
    PairDev* pairs = nullptr;
    PairDev* pairs_unit = nullptr;
    TileDev* tiles = nullptr;
    uint32_t* wave_tiles = nullptr;
    double* pair_u0 = nullptr;  // pass mode: unit baselines at the starting point (host_pair_u0)
    double* m0 = nullptr;
    double* m1 = nullptr;
  } t;
  char* h_arena = nullptr;
  size_t h_arena_cap = 0;
  // mvo_hip_staging: pinned buffers the caller's gather loop writes into (offsets, m0, m1)
  uint64_t* h_st_offs = nullptr;
  double* h_st_m0 = nullptr;
  double* h_st_m1 = nullptr;
  uint32_t st_pairs = 0;      // capacity in pairs (+1 offsets)
  uint64_t st_blocks = 0;     // capacity in residual blocks
  bool copy_pending = false;  // an async copy out of h_arena may still be in flight
  // experiment switches, read from the environment ONCE, at mvo_hip_create (a getenv is a walk through
  // the whole environment: four of them were ~1 us of a 33-us default-size refine)
  uint32_t env_waves_per_cu = 16;
  int env_overlap_chunks = 0;  // 0: the default (kOverlapChunks)
  bool env_no_host_results = false, env_no_poll = false;
  uint32_t env_host_results_max_w_loss = 1024;  // ... under the loss (the same variable sets both)
  uint32_t env_host_results_max_w = 1024;  // MVO_HIP_HOST_RESULTS_MAX_W
  uint32_t env_slots_min_w = 1, env_slots_max_p = 64;  // MVO_HIP_SLOTS_MIN_W / _MAX_P (experiments): slots plan for W > / P <=
  bool one_shot = false;        // set by mvo_hip_refine* around their upload: the upload is solved exactly once
  int env_fused_max_blocks = 256;  // MVO_HIP_FUSED_MAX_BLOCKS: fold workgroups up to which fold and LM loop are one launch
  bool env_arena_copy = false;  // MVO_HIP_ARENA_COPY: a small window's arena is copied to the device like any other
  bool env_lm4_sums = false;  // MVO_HIP_LM4_SUMS: k_lm4 behind a fold WITH ticket tails, cut as for the other kernels (round 3's plan)
  bool env_no_tails = false;  // MVO_HIP_NO_TAILS: per-pair sums by k_reduce_slots launches instead of the passes' ticket tails
  bool env_lm_generic = false;  // MVO_HIP_LM_GENERIC: k_lm<NT, 0> for every solve (no compile-time specialisation)
  bool env_no_persistent = false;  // MVO_HIP_NO_PERSISTENT: the robust path launch by launch, never k_lm<256, 6>
  bool env_persistent_test_fail = false;
  bool persistent_failed = false;  // a persistent launch ended without a result once: not tried again on this handle
  bool persistent_in_flight = false;
  // fold + LM loop as one launch (k_lm<256, 7>): MVO_HIP_NO_FUSED keeps them apart; so does a handle on which such a
  // launch once ended without a result.  fused_count: what the device's count of finished pairs reads before a launch
  // (it is never reset on the device; d_grid word 16, error word 48 -- words 0 / 32 belong to the persistent solver,
  // whose launches clear the whole block)
  bool env_no_fused = false, env_fused_test_fail = false;
  bool fused_failed = false, fused_in_flight = false;
  uint32_t fused_count = 0;
  bool fused_ready = false;  // d_grid exists, holds fused_count and a zeroed counter for the next persistent launch
  uint32_t grid_seq = 0;     // persistent launches on this handle since d_grid was last cleared
  int n_cus = 0;
  // d_ctl = [window status: W int32, padded][done counter: 4 int32][results: W LmResult];
  // the last two are mirrored in pinned host memory by one copy
  int32_t* d_status = nullptr;
  int32_t* d_done = nullptr;
  LmResult* d_results = nullptr;
  bool slots_plan = false;     // this upload: the LM kernels read the fold's slots, no ticket tails (mvo_hip_upload)
  bool sums_external = false;  // the per-pair sums in d_pairsum were put there by the caller (fused track fold)
  bool results_on_host = false;  // this solve's k_lm writes its (few) result records straight into h_results
  char* h_ctl = nullptr;  // pinned: [done: 4 int32][results]
  size_t h_ctl_cap = 0;
  int32_t* h_done = nullptr;
  LmResult* h_results = nullptr;
  // host-side staging vectors, kept between calls so that a steady-state upload allocates nothing
  std::vector<WinDev> v_wins;
  std::vector<PairDev> v_unit;
  std::vector<TileDev> v_tiles;
  std::vector<uint32_t> v_wave_tiles;
  std::vector<double> v_initx;
  std::vector<int32_t> v_clamped;  // per window: bit 2k = scale_k clamped low, bit 2k+1 = high (IR.cpp:350-363)
  bool lm_started = false;  // false until the first k_lm launch of the current solve
  bool lm4_marked = false;  // the current solve is one k_lm4 launch whose completion is read off the records' marks
  int32_t lm4_mark = 0;
  uint32_t lm4_seq = 0;
  // The done counter is never cleared between solves: a solve is complete when it has advanced by W
  // past done_base (one fill less per solve).  done_ready = the counter holds done_base.
  uint32_t done_base = 0;
  bool done_ready = false;
  bool profiling = false;  // record HIP events around the launches (mvo_hip_set_profiling)
  // problem description of the last upload
  bool uploaded = false;
  bool solved = false;  // h_results hold the results of a completed solve of the current upload
  uint32_t W = 0, M = 0, N = 0, P = 0;
  uint64_t n_obs = 0;      // residual blocks (all windows)
  uint64_t c_elems = 0;    // padded plane length
  uint32_t n_tiles = 0, n_slots = 0;
  uint32_t n_waves = 0;          // wavefronts of an observation pass (one chunk each)
  uint32_t max_block_tiles = 0;  // LDS rows a workgroup needs for its tiles' sums
  mvo_hip_options opt{};
  bool pass_mode = false;
  bool inline_eval = false;  // k_lm runs the observation passes itself (small or many windows)
  bool tickets_clean = false;  // d_pticket is all zero (it is after every completed launch)
  bool comm_broken = false;      // an ncclAllReduce call failed on this communicator
  bool pairsum_foreign = false;  // d_pairsum holds all-reduced sums (partitioned no-loss solve): rows of pairs that are empty HERE are not zero
  std::vector<mvo_hip_frame> h_frames;  // copy of the uploaded frames (write-back base)
  std::vector<PairDev> h_pairs;
  // trace
  int32_t trace_cap = 0;
  // timing
  hipEvent_t ev[2 * kMaxTimedLaunches + 2] = {};
  bool events_ready = false;
  mvo_hip_timing timing{};
  // batched windows without a loss: the fold of chunk c + 1 (HBM-bound) runs on `stream` while the LM
  // loops of the windows chunk c completed (issue-bound) run on `stream_lm`
  hipStream_t stream_lm[3] = {nullptr, nullptr, nullptr};  // consecutive chunks' LM loops overlap too (4 hardware queues with `stream`)
  hipEvent_t ev_chunk[kMaxOverlapChunks + 3] = {};
  bool overlap_ready = false;
  int n_chunks = 0;                                      // 0: no overlap for this upload
  uint32_t chunk_block_end[kMaxOverlapChunks] = {};      // fold workgroups [.., end)
  uint32_t chunk_win_end[kMaxOverlapChunks] = {};        // windows whose tiles all lie in workgroups < end
  int chunk_kind[kMaxOverlapChunks] = {};                // 0: by shape (k_lm<64> / k_lm<256>); 4: k_lm4; 64 / 256: that kernel
  bool pipelined = false;                                // the plan below is the staggered one (batch_pipeline_plan)
  std::vector<uint32_t> v_win_last_wave;
  // observation-partitioned solve: this rank's RCCL communicator (mvo_hip_comm_init); collectives are
  // queued on `stream` between the observation pass and the LM step
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  ncclResult_t last_rccl_error = ncclSuccess;
};

namespace {

double host_scale(double t, double max_len, double min_len) {
  double e = std::exp(-1.0 * t);
  if (std::isinf(e)) return min_len;
  return min_len + ((max_len - min_len) / (1.0 + e));
}

// IR.cpp:348-368
// The reference recurses on the clamped value; one clamp step lands strictly inside the interval
// whenever the interval is wider than a few ulps of its ends (validate() demands that), so the
// recursion is at most one level deep -- written as a bounded loop so that no input can overflow the
// stack.  *clamped: 0 none, 1 low (IR.cpp:350-357), 2 high (IR.cpp:358-363) -- the reference warns.
double host_reverse_scale(double length, double max_len, double min_len, int* clamped = nullptr) {
  const double eps = std::numeric_limits<double>::epsilon();
  for (int depth = 0; depth < 4; ++depth) {
    if (length <= min_len) {
      if (clamped && !*clamped) *clamped = 1;
      length = min_len + (length * eps) + eps;
    } else if (length >= max_len) {
      if (clamped && !*clamped) *clamped = 2;
      length = max_len - (length * eps + eps);
    } else {
      break;
    }
  }
  return -1.0 * std::log((max_len - length) / (length - min_len));
}

uint32_t num_pairs(uint32_t M, uint32_t N) { return M * (N - 1) - (M * (M - 1)) / 2; }

int validate(uint32_t n_windows, const void* frames, uint32_t M, uint32_t N, const mvo_hip_options* opt,
             const mvo_hip_observations* obs) {
  if (!frames || !opt || !obs || n_windows == 0) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (M == 0 || N <= M) return MVO_HIP_ERR_INVALID_ARGUMENT;  // IR.cpp:29-30
  if (M > MVO_HIP_MAX_REFINE) return MVO_HIP_ERR_UNSUPPORTED; // IR.cpp:450
  if (N > MVO_HIP_MAX_FRAMES) return MVO_HIP_ERR_UNSUPPORTED;
  if (opt->max_num_iterations < 0) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (!(opt->highest_length > opt->lowest_length) || !std::isfinite(opt->highest_length) ||
      !std::isfinite(opt->lowest_length))
    return MVO_HIP_ERR_INVALID_ARGUMENT;
  {
    // reverseScale's clamp (IR.cpp:350-363) must land strictly inside (lowest, highest); the
    // reference would recurse forever otherwise (e.g. lowest 4, highest 4 + 1 ulp)
    const double eps = std::numeric_limits<double>::epsilon();
    const double lo = opt->lowest_length + (opt->lowest_length * eps) + eps;
    const double hi = opt->highest_length - (opt->highest_length * eps + eps);
    if (!(lo > opt->lowest_length && lo < opt->highest_length && hi < opt->highest_length && hi > opt->lowest_length))
      return MVO_HIP_ERR_INVALID_ARGUMENT;
    const mvo_hip_frame* fr = static_cast<const mvo_hip_frame*>(frames);
    for (uint32_t w = 0; w < n_windows; ++w)
      for (uint32_t k = 0; k < M; ++k)
        if (std::isinf(fr[(size_t)w * N + k].scale)) return MVO_HIP_ERR_INVALID_ARGUMENT;  // the reference would recurse forever
  }
  const uint32_t P = num_pairs(M, N);
  uint64_t total = 0;
  for (uint32_t w = 0; w < n_windows; ++w) {
    const mvo_hip_observations& o = obs[w];
    if (o.n_pairs != P || !o.pair_offsets) return MVO_HIP_ERR_INVALID_ARGUMENT;
    if (o.pair_offsets[0] != 0) return MVO_HIP_ERR_INVALID_ARGUMENT;
    for (uint32_t p = 0; p < P; ++p)
      if (o.pair_offsets[p + 1] < o.pair_offsets[p]) return MVO_HIP_ERR_INVALID_ARGUMENT;
    if (o.pair_offsets[P] > 0 && (!o.m0 || !o.m1)) return MVO_HIP_ERR_INVALID_ARGUMENT;
    total += o.pair_offsets[P];
  }
  if (total + 64ull * P * n_windows >= (1ull << 31)) return MVO_HIP_ERR_UNSUPPORTED;
  return MVO_HIP_OK;
}

// one window of at most MVO_SMALL_PAIRS frame pairs, no loss, that folds its own observations (the reference's shipped window)
bool small_window(const mvo_hip_context* h) {
  return h->inline_eval && !h->pass_mode && h->W == 1 && h->P <= MVO_SMALL_PAIRS && !h->env_lm_generic;
}
SolveParams make_params(const mvo_hip_context* h) {
  SolveParams prm{};
  prm.ftol = h->opt.function_tolerance;
  prm.gtol = h->opt.gradient_tolerance;
  prm.ptol = h->opt.parameter_tolerance;
  prm.min_len = h->opt.lowest_length;
  prm.max_len = h->opt.highest_length;
  prm.max_it = h->opt.max_num_iterations;
  prm.use_loss = h->opt.use_loss_function ? 1 : 0;
  prm.fix_length = h->opt.fix_length ? 1 : 0;
  prm.pass_mode = h->pass_mode ? 1 : 0;
  prm.trace_cap = h->trace_cap;
  prm.inline_eval = h->inline_eval ? 1 : 0;
  prm.M = (int32_t)h->M;
  prm.N = (int32_t)h->N;
  prm.P = (int32_t)h->P;
  // one small window that evaluates its own observations: where each pair's rays lie goes with the kernel arguments
  prm.small_tab = 0;
  if (small_window(h) && h->h_pairs.size() == h->P) {
    prm.small_tab = 1;
    for (uint32_t p = 0; p < h->P; ++p) prm.small_offs[p] = h->h_pairs[p].src_start;
    prm.small_offs[h->P] = h->P ? h->h_pairs[h->P - 1].src_start + h->h_pairs[h->P - 1].count : 0;
  }
  return prm;
}

int ensure_events(mvo_hip_context* h) {
  if (h->events_ready) return MVO_HIP_OK;
  for (auto& e : h->ev) HIP_TRY(hipEventCreate(&e));
  h->events_ready = true;
  return MVO_HIP_OK;
}

bool lm4_eligible(const mvo_hip_context* h);
bool slots_plan_static(const mvo_hip_context* h);

// Unit baselines of window w's pairs at the starting point (v_initx): what k_lm publishes for a
// candidate (pair_unit_baseline, IR.cpp:137-144, 254-255, 298-309, 466-577), here on the host so that the
// fold can be the evaluation pass at x0 as well.  out: 4 doubles per pair (u, |baseline|).
void host_pair_u0(const mvo_hip_context* h, uint32_t w, double* out) {
  const uint32_t M = h->M, N = h->N;
  const double max_len = h->opt.highest_length, min_len = h->opt.lowest_length;
  double sv[MVO_MAXM][3];
  for (uint32_t k = 0; k < M; ++k) {
    const double* x = &h->v_initx[((size_t)w * MVO_MAXM + k) * 4];
    const double e = std::exp(-1.0 * x[3]);  // scaleTemplated, IR.cpp:298-309
    const double sk = std::isinf(e) ? min_len : min_len + ((max_len - min_len) / (1.0 + e));
    for (int j = 0; j < 3; ++j) sv[k][j] = sk * x[j];
  }
  uint32_t p = 0;
  for (uint32_t a = 0; a < M; ++a)
    for (uint32_t b = a; b < N - 1; ++b, ++p) {
      const uint32_t kmax = b < M - 1 ? b : M - 1;
      double bx = sv[a][0], by = sv[a][1], bz = sv[a][2];
      for (uint32_t k = a + 1; k <= kmax; ++k) {
        bx += sv[k][0];
        by += sv[k][1];
        bz += sv[k][2];
      }
      if (b >= M) {  // IR.cpp:137-144: only frame b itself
        const mvo_hip_frame& f = h->h_frames[(size_t)w * N + b];
        bx += f.scale * f.vec[0];
        by += f.scale * f.vec[1];
        bz += f.scale * f.vec[2];
      }
      const double nb = std::sqrt(bx * bx + by * by + bz * bz);
      out[4 * p + 0] = bx / nb;
      out[4 * p + 1] = by / nb;
      out[4 * p + 2] = bz / nb;
      out[4 * p + 3] = nb;
    }
}

// rays_on_device: obs[w].m0 / m1 are not read; the caller fills h->t.m0 / m1 itself with work it queues
// on h->stream after this call (the device-side gather of mvo_tracks.hip); pair_offsets are host memory.
int do_upload(mvo_hip_context* h, uint32_t W, const mvo_hip_frame* frames, uint32_t M, uint32_t N,
              const mvo_hip_options* opt, const mvo_hip_observations* obs, bool rays_on_device = false,
              bool no_rays = false, bool rays_pinned = false) {
  const auto t_begin = std::chrono::steady_clock::now();
  int rc = validate(W, frames, M, N, opt, obs);
  if (rc != MVO_HIP_OK) return rc;
  HIP_TRY(hipSetDevice(h->device));
  h->uploaded = false;
  h->solved = false;
  h->lm4_marked = false;
  h->sums_external = false;
  const uint32_t P = num_pairs(M, N);
  h->W = W;
  h->M = M;
  h->N = N;
  h->P = P;
  h->opt = *opt;
  h->pass_mode = opt->use_loss_function || (opt->assembly_mode & MVO_HIP_ASSEMBLY_PASSES);

  uint64_t n_obs = 0;
  for (uint32_t w = 0; w < W; ++w) n_obs += obs[w].pair_offsets[P];
  h->n_obs = n_obs;
  // Who runs the observation passes?  For small windows one workgroup is faster than the launches
  // it saves: each window's k_lm workgroup evaluates its own observations and the whole solve is
  // ONE launch.  Larger windows use the chip-wide tiled passes (measured: 4096 windows of 10.5 k
  // residual blocks run 5 % faster tiled).  MVO_HIP_ASSEMBLY_TILED forces the latter.
  {
    uint64_t max_win_obs = 0;
    for (uint32_t w = 0; w < W; ++w)
      if (obs[w].pair_offsets[P] > max_win_obs) max_win_obs = obs[w].pair_offsets[P];
    // ... except for batches that k_lm4 takes four windows per wavefront: one chip-wide fold and a
    // quarter of the LM wavefronts beat a fold per window inside its own workgroup
    h->inline_eval = false;
    const bool four_per_wave = lm4_eligible(h);
    // One window without a loss that the fused launch can take (fold chip-wide and LM loop in ONE launch, fused_ok):
    // that launch is ahead from ~3 k residual blocks on (N = 10, M = 5, fixed 10 iterations, no event records: 2100 blocks
    // 51.1 us inline against 52.6, 4200: 53.8 against 51.0, 8050: 56.4 against 50.3; 135 pairs, 8100 blocks: 67.8 against
    // 54.0 -- scripts/exp/inline_vs_fused.py).  (Event records asked for AFTER the upload make it two launches.)
    const bool fused_likely = W == 1 && !h->pass_mode && !opt->fix_length && h->trace_cap == 0 && N - 1 - M <= 32 &&
                              !h->env_no_fused && !h->fused_failed && !h->env_lm_generic && !h->env_no_tails &&
                              !h->profiling && n_obs >= 16ull * P;
    // Batches without a loss of windows with at most 64 pairs (the slots plan's domain, below): the chip-wide fold without
    // tails is level with the inline fold or ahead from ~3 k blocks per window on (scripts/exp/batch_inline_vs_tiled.py,
    // call time inline / tiled: 64 windows of 1050 / 3500 / 7000 blocks 54.2 / 60.3, 57.3 / 54.4, 60.0 / 53.7 us), and beyond
    // one window per CU the inline fold is one wavefront per window: 512 windows 106.8 / 102.8, 129.2 / 103.4, 145.7 / 112.3.
    // One window UNDER THE LOSS that the persistent robust solver can take (robust_persistent_ok): the inline pass is one
    // wavefront per PAIR and evaluation, the persistent solver a workgroup per 1024 blocks -- scripts/exp/
    // inline_vs_persistent_loss.py, fixed 10 iterations, inline / persistent, us per solve: 3 pairs, 180 blocks 59.9 / 101.0;
    // 7, 1400: 95.5 / 116.0; 11, 3300: 113.6 / 106.3; 18, 1080: 122.8 / 137.3; 18, 5400: 174.5 / 121.2; 22, 6600: 200.9 /
    // 133.7; 30, 1800: 165.4 / 148.8; 35, 1050: 181.2 / 175.5; 35, 8050: 275.5 / 142.7; 85, 5100: 360.8 / 162.0; 135, 8100:
    // 545.6 / 171.4.  Inline stays for few pairs or few blocks.
    const bool persistent_likely = W == 1 && opt->use_loss_function && !opt->fix_length && h->trace_cap == 0 &&
                                   !h->env_no_persistent && !h->persistent_failed && !h->env_lm_generic &&
                                   !h->env_no_tails && P <= (uint32_t)kRobustMaxPairs;
    const bool persistent_ahead = persistent_likely && ((P >= 10 && n_obs >= 2048) || (P >= 28 && n_obs >= 1536));
    const bool slots_batch = W > 1 && !h->pass_mode && P <= 64;
    // Batches under the loss (an observation pass per evaluation): inline passes are one wavefront per pair and
    // evaluation inside the window's workgroup -- or inside its ONE wavefront beyond a window per CU
    // (scripts/exp/batch_inline_vs_tiled_loss.py, call time inline / tiled: 64 windows of 1050 / 3500 blocks 186.9 / 283.9,
    // 233.6 / 204.8 us; 256 windows 195.3 / 331.7, 300.6 / 288.5; 512 windows 610.1 / 423.7, 962.0 / 385.8; 1024 windows
    // 682.5 / 525.3, 1014.9 / 549.5).
    const bool pass_batch = W > 1 && h->pass_mode;
    const uint64_t inline_limit = fused_likely ? kInlineFusedLimit
                                  : slots_batch ? (W > 256 ? 0 : kInlineFusedLimit)
                                  : pass_batch ? (W > 256 ? 0 : kInlinePassBatchLimit) : kInlineWindowLimit;
    h->inline_eval = max_win_obs <= inline_limit && !(opt->assembly_mode & MVO_HIP_ASSEMBLY_TILED) && !four_per_wave &&
                     !persistent_ahead;
  }
  // One chunk of consecutive residual blocks per wavefront, all wavefronts resident at once:
  // waves_per_cu * 256 CUs of them, but at least kMinChunk blocks each so that small problems
  // do not pay one reduction per handful of blocks.
  const uint32_t waves_per_cu = h->env_waves_per_cu;  // MVO_HIP_WAVES_PER_CU, tuning knob for experiments
  const uint64_t kMinChunk = 256, kMaxWaveTiles = 255;
  // MVO_HIP_OVERLAP=1 (off by default since round 3): batched windows without a loss run the fold in
  // kMaxOverlapChunks launches next to the LM loops of the chunks before (do_solve); every one of those launches
  // must fill the chip by itself.  It paid in round 1, when a window's LM loop took four times what it takes now;
  // measured again at the end of round 3 (scripts/exp/lm4_crossover.py, config-3 windows, device time per batch):
  // 512 / 768 / 1024 windows 200 / 253 / 305 us overlapped against 118 / 156 / 189 us fold-then-LM -- the LM
  // wavefronts own their SIMDs' registers and push the fold's wavefronts off them, and the chunked fold loses
  // bandwidth.  Kept as an experiment switch with its test.
  const int overlap_chunks = h->env_overlap_chunks ? h->env_overlap_chunks : kOverlapChunks;  // MVO_HIP_OVERLAP_CHUNKS
  const bool want_overlap = !h->pass_mode && !h->inline_eval && W >= 256 && P <= 256 && !lm4_eligible(h) &&
                            n_obs >= (uint64_t)kMinChunk * 256 * waves_per_cu * overlap_chunks &&
                            std::getenv("MVO_HIP_OVERLAP") != nullptr && !std::getenv("MVO_HIP_NO_OVERLAP");
  // Batches that k_lm4 takes (thousands of windows of <= 64 pairs): the fold runs in four launches and the
  // LM loops of the windows already folded start beside it -- STAGGERED by kernel: the bulk on k_lm4 (four
  // windows per wavefront, ~160 us per launch whatever its size), the windows folded too late for that on
  // k_lm<64> (~120 us), the last two times 256 on k_lm<256> (~65 us, one window per CU), so that what is
  // left after the fold's end is the shortest LM latency instead of the longest (do_solve).
  // MEASURED SLOWER than fold-then-k_lm4 (config 3: 0.75-0.82 ms against 0.67-0.70, profiles/r3/NOTES.md): the
  // LM wavefronts beside the fold take its issue slots and the 16 x finer tiling costs it bandwidth (5.1 ->
  // 4.2-4.7 TB/s), more than the 90 us of k_lm4 latency the stagger hides.  Kept as an experiment switch
  // (MVO_HIP_PIPELINE), off by default, with its test.
  const bool want_pipeline = !h->pass_mode && !h->inline_eval && lm4_eligible(h) && W >= 2048 && P <= 64 &&
                             !(opt->assembly_mode & (MVO_HIP_ASSEMBLY_ONE_PER_WAVE | MVO_HIP_ASSEMBLY_FOUR_PER_WAVE)) &&
                             n_obs >= 16000000ull && std::getenv("MVO_HIP_PIPELINE") != nullptr;
  // Batches of windows with at most 64 pairs (slots_plan): k_lm4 / k_lm<64> add the slots of their pairs up themselves
  // and the fold runs WITHOUT its ticket tails.  Without them it is the faster the finer it is cut -- many rounds of short
  // wavefronts that fill in behind one another instead of one round that ends with its slowest wavefront (config 3,
  // k_fold per launch: 386 us with tails at any granularity; without: 370 us at 16 wavefronts per CU, 364 / 352 / 347 /
  // 349 / 345 us at 128 / 256 / 384 / 512 / 1024, scripts/exp/env_kstats.sh).  With tails (every other plan) finer is
  // SLOWER: config 4 84 -> 92 / 99 us at 32 / 128.  An LM lane adds its pair's slots one after the other, so the cut
  // stays coarse enough for a pair to have a few of them (a workgroup's share >= half the mean pair, 1/8 of the longest);
  // windows so long that this would leave the chip short of wavefronts keep the tails.
  const uint64_t chunk_tails_plan = std::max<uint64_t>(kMinChunk, (n_obs + 256ull * waves_per_cu - 1) / (256ull * waves_per_cu));
  uint64_t chunk_slots_plan = 0;
  h->slots_plan = false;
  if (slots_plan_static(h) && !want_pipeline && n_obs > 0) {
    uint64_t max_pair = 0;
    for (uint32_t w = 0; w < W; ++w)
      for (uint32_t q = 0; q < P; ++q) max_pair = std::max<uint64_t>(max_pair, obs[w].pair_offsets[q + 1] - obs[w].pair_offsets[q]);
    const uint64_t mean_pair = n_obs / ((uint64_t)W * P);
    chunk_slots_plan = std::max<uint64_t>({kMinChunk, (n_obs + 256ull * kLm4FoldWavesPerCu - 1) / (256ull * kLm4FoldWavesPerCu),
                                           mean_pair / (2 * FOLD_WAVES), max_pair / (8 * FOLD_WAVES)});
    h->slots_plan = chunk_slots_plan <= chunk_tails_plan;
  }
  // (every launch of a chunked fold has to fill the chip by itself; the staggered plan's last launches cover 1/16 of the windows)
  const uint64_t max_waves = h->slots_plan ? (n_obs + chunk_slots_plan - 1) / chunk_slots_plan
                                           : 256ull * waves_per_cu * (want_pipeline ? 16 : want_overlap ? overlap_chunks : 1);
  uint64_t want_waves = (n_obs + kMinChunk - 1) / kMinChunk;
  if (want_waves > max_waves) want_waves = max_waves;
  if (want_waves == 0) want_waves = 1;
  const uint64_t chunk = (n_obs + want_waves - 1) / want_waves > 0 ? (n_obs + want_waves - 1) / want_waves : 1;

  std::vector<WinDev>& wins = h->v_wins;
  wins.assign(W, WinDev{});
  std::vector<PairDev>& pairs = h->h_pairs;
  pairs.assign((size_t)W * P, PairDev{});
  std::vector<TileDev>& tiles = h->v_tiles;
  tiles.clear();
  tiles.reserve((size_t)want_waves + (size_t)W * P + 8);
  std::vector<uint32_t>& wave_tiles = h->v_wave_tiles;  // first tile of every wavefront, plus the end
  wave_tiles.clear();
  wave_tiles.reserve((size_t)want_waves + 8);
  wave_tiles.push_back(0);
  uint64_t wave_fill = 0;  // residual blocks already given to the current wavefront
  std::vector<double>& initx = h->v_initx;
  initx.assign((size_t)W * MVO_MAXM * 4, 0.0);
  h->v_clamped.assign(W, 0);
  uint64_t src = 0, cpos = 0;
  uint32_t slot = 0;
  std::vector<uint32_t>& win_last_wave = h->v_win_last_wave;
  win_last_wave.assign(W, 0);
  for (uint32_t w = 0; w < W; ++w) {
    win_last_wave[w] = w ? win_last_wave[w - 1] : 0;  // a window without observations waits for nothing new
    wins[w].frame_off = w * N;
    wins[w].pair_off = w * P;
    wins[w].N = N;
    wins[w].M = M;
    wins[w].P = P;
    for (uint32_t k = 0; k < M; ++k) {
      const mvo_hip_frame& f = frames[(size_t)w * N + k];
      double* x = &initx[((size_t)w * MVO_MAXM + k) * 4];
      x[0] = f.vec[0];
      x[1] = f.vec[1];
      x[2] = f.vec[2];
      int clamped = 0;
      x[3] = host_reverse_scale(f.scale, opt->highest_length, opt->lowest_length, &clamped);  // IR.cpp:102
      if (clamped) h->v_clamped[w] |= 1 << (2 * k + (clamped - 1));
    }
    uint32_t p = 0;
    for (uint32_t a = 0; a < M; ++a)
      for (uint32_t b = a; b < N - 1; ++b, ++p) {
        PairDev& pd = pairs[(size_t)w * P + p];
        pd.a = a;
        pd.b = b;
        pd.src_start = (uint32_t)src;
        pd.count = (uint32_t)(obs[w].pair_offsets[p + 1] - obs[w].pair_offsets[p]);
        const uint64_t cnt = obs[w].pair_offsets[p + 1] - obs[w].pair_offsets[p];
        pd.slot_begin = slot;
        uint32_t cur_slot = 0;
        for (uint64_t done = 0; done < cnt;) {
          // close the current wavefront when its chunk is full (or its tile list is)
          if (wave_fill >= chunk || tiles.size() - wave_tiles.back() >= kMaxWaveTiles) {
            wave_tiles.push_back((uint32_t)tiles.size());
            wave_fill = 0;
          }
          const uint64_t take = (cnt - done) < (chunk - wave_fill) ? (cnt - done) : (chunk - wave_fill);
          TileDev t{};
          t.pair = w * P + p;
          t.win = w;
          t.src_start = (uint32_t)(src + done);
          t.c_start = (uint32_t)(cpos + done);
          t.count = (uint32_t)take;
          t.f0 = w * N + a;      // IR.cpp:172  R0 = frames[frame0].R
          t.f1 = w * N + b + 1;  // IR.cpp:171  R1 = frames[frame1 + 1].R
          // a slot never spans workgroups: new slot for a new pair or at a workgroup boundary
          const bool block_start = tiles.size() == wave_tiles.back() && ((wave_tiles.size() - 1) % FOLD_WAVES) == 0;
          if (done == 0 || block_start) cur_slot = slot++;
          t.slot = cur_slot;
          tiles.push_back(t);
          win_last_wave[w] = (uint32_t)wave_tiles.size() - 1;  // index of the wavefront this tile belongs to
          done += take;
          wave_fill += take;
        }
        pd.slot_end = slot;
        src += cnt;
        cpos += (cnt + 31) & ~31ull;  // pair segments of c start on 256-byte boundaries
      }
  }
  if (tiles.size() > wave_tiles.back()) wave_tiles.push_back((uint32_t)tiles.size());
  h->n_tiles = (uint32_t)tiles.size();
  h->n_waves = (uint32_t)wave_tiles.size() - 1;
  h->max_block_tiles = 1;
  for (uint32_t g = 0; g < h->n_waves; g += FOLD_WAVES) {
    const uint32_t e = g + FOLD_WAVES < h->n_waves ? g + FOLD_WAVES : h->n_waves;
    const uint32_t nt = wave_tiles[e] - wave_tiles[g];
    if (nt > h->max_block_tiles) h->max_block_tiles = nt;
  }
  h->n_slots = slot;
  h->c_elems = cpos + 32;

  // overlap plan (no loss, many windows, LM loop in its own launch): equal ranges of fold workgroups;
  // a window may start its LM loop once the last workgroup holding one of its tiles is done
  h->n_chunks = 0;
  h->pipelined = false;
  if (want_pipeline) {
    // fold time per window at ~5 TB/s (48 B per residual block), in microseconds
    const double us_per_win = 48.0 * (double)n_obs / (double)W / 5.1e6;
    uint32_t tail = 256;                                         // k_lm<256>: one window per CU
    uint32_t mid = (uint32_t)std::lround(45.0 / us_per_win);     // what the fold passes in the ~45 us k_lm<64> is ahead of k_lm4
    if (const char* e = std::getenv("MVO_HIP_PIPELINE_MID")) mid = (uint32_t)std::atoi(e);
    if (mid < 128) mid = 128;
    if (mid > 768) mid = 768;
    const uint32_t ends[4] = {W - 2 * tail - mid, W - 2 * tail, W - tail, W};
    const int kinds[4] = {4, 64, 256, 256};
    for (int c = 0; c < 4; ++c) {
      h->chunk_win_end[c] = ends[c];
      h->chunk_kind[c] = kinds[c];
      // fold workgroups up to the last one that holds a tile of these windows
      h->chunk_block_end[c] = c == 3 ? (h->n_waves + FOLD_WAVES - 1) / FOLD_WAVES : win_last_wave[ends[c] - 1] / FOLD_WAVES + 1;
    }
    // (a workgroup that straddles a chunk boundary belongs to the earlier fold launch: windows of the later
    // chunk that have tiles in it simply find them done)
    h->n_chunks = 4;
    h->pipelined = true;
  } else {
    const uint32_t blocks = (h->n_waves + FOLD_WAVES - 1) / FOLD_WAVES;
    if (want_overlap && blocks >= 256u * (uint32_t)overlap_chunks) {
      h->n_chunks = overlap_chunks;
      uint32_t w = 0;
      for (int c = 0; c < h->n_chunks; ++c) {
        const uint32_t end = c + 1 == h->n_chunks ? blocks : (uint32_t)((uint64_t)blocks * (c + 1) / h->n_chunks);
        h->chunk_block_end[c] = end;
        while (w < W && win_last_wave[w] / FOLD_WAVES < end) ++w;
        h->chunk_win_end[c] = c + 1 == h->n_chunks ? W : w;
      }
    }
  }

  h->h_frames.assign(frames, frames + (size_t)W * N);
  std::vector<PairDev>& unit = h->v_unit;  // the same pairs, each reading its own row of d_pairsum
  unit.assign(pairs.begin(), pairs.end());
  for (size_t q = 0; q < unit.size(); ++q) {
    unit[q].slot_begin = (uint32_t)q;
    unit[q].slot_end = (uint32_t)q + 1;
  }

  // ---- lay the tables out in the arena; rays join them when they are few (one copy in all)
  // (rays already in the handle's pinned staging block and a window folded chip-wide: the fold reads them there --
  // zero_copy below --, a gather into the arena would be one more pass over them)
  const bool rays_in_arena = n_obs <= kArenaRayLimit && !no_rays &&
                             !(rays_pinned && W == 1 && !h->inline_eval && !std::getenv("MVO_HIP_STAGED_COPY"));
  size_t off = 0;
  auto place = [&off](size_t bytes) {
    const size_t o = off;
    off = (off + bytes + 255) & ~(size_t)255;
    return o;
  };
  const size_t o_frames = place(sizeof(mvo_hip_frame) * (size_t)W * N);
  const size_t o_initx = place(sizeof(double) * initx.size());
  const size_t o_wins = place(sizeof(WinDev) * W);
  const size_t o_pairs = place(sizeof(PairDev) * pairs.size());
  const size_t o_unit = place(sizeof(PairDev) * unit.size());
  const size_t o_tiles = place(sizeof(TileDev) * (tiles.size() + 1));
  const size_t o_wave = place(sizeof(uint32_t) * (wave_tiles.size() + 1));
  const size_t o_pu0 = place(h->pass_mode ? sizeof(double) * 4 * pairs.size() : 0);
  const size_t o_m0 = rays_in_arena ? place(sizeof(double) * 3 * (n_obs + 1)) : 0;
  const size_t o_m1 = rays_in_arena ? place(sizeof(double) * 3 * (n_obs + 1)) : 0;
  const size_t arena_bytes = off;

  hipStream_t s = h->stream;
  if (h->copy_pending) {  // the previous upload may still be reading the staging block
    HIP_TRY(hipStreamSynchronize(s));
    h->copy_pending = false;
  }
  if (h->h_arena_cap < arena_bytes) {
    if (h->h_arena) (void)hipHostFree(h->h_arena);
    h->h_arena = nullptr;
    h->h_arena_cap = 0;
    const size_t want = arena_bytes + arena_bytes / 4 + 4096;
    HIP_TRY(hipHostMalloc((void**)&h->h_arena, want, hipHostMallocDefault));
    h->h_arena_cap = want;
  }
  HIP_TRY(h->d_arena.reserve(arena_bytes));
  if (!rays_in_arena && !no_rays && !(rays_pinned && W == 1 && !std::getenv("MVO_HIP_STAGED_COPY"))) {
    HIP_TRY(h->d_m0.reserve(sizeof(double) * 3 * (n_obs + 1)));
    HIP_TRY(h->d_m1.reserve(sizeof(double) * 3 * (n_obs + 1)));
  }
  if (h->pass_mode) HIP_TRY(h->d_c.reserve(sizeof(double) * 3 * h->c_elems));
  HIP_TRY(h->d_partial.reserve(sizeof(double) * 8 * ((size_t)slot + 1)));
  HIP_TRY(h->d_pair_u.reserve(sizeof(double) * 4 * pairs.size()));
  HIP_TRY(h->d_pairsum.reserve(sizeof(double) * 8 * pairs.size()));
  {
    // one ticket per pair for the passes' tails; every completed launch leaves them at zero
    const size_t cap_before = h->d_pticket.cap;  // (not the pointer: a new block may get the old address)
    HIP_TRY(h->d_pticket.reserve(sizeof(int32_t) * pairs.size()));
    if (h->d_pticket.cap != cap_before) h->tickets_clean = false;
    // a pair without residual blocks has no slot and no tail: its sums are zero from here on
    bool empty_pair = false;
    for (const PairDev& pd : pairs) empty_pair |= pd.count == 0;
    if (empty_pair) HIP_TRY(hipMemsetAsync(h->d_pairsum.p, 0, sizeof(double) * 8 * pairs.size(), h->stream));
    if (h->pass_mode) {
      HIP_TRY(h->d_proj.reserve(sizeof(double) * 10 * pairs.size() * 2));
      // a pair without residual blocks is never projected: zeros from here on (both halves)
      if (empty_pair) HIP_TRY(hipMemsetAsync(h->d_proj.p, 0, sizeof(double) * 10 * pairs.size() * 2, h->stream));
    }
    h->pairsum_foreign = false;
  }
  HIP_TRY(h->d_states.reserve(sizeof(LmState) * W));
  {
    const size_t status_bytes = (sizeof(int32_t) * W + 15) & ~(size_t)15;
    const size_t mirror_bytes = 16 + sizeof(LmResult) * W;
    const size_t ctl_cap_before = h->d_ctl.cap;  // (not the pointer: a new block may get the old address)
    HIP_TRY(h->d_ctl.reserve(status_bytes + mirror_bytes));
    if (h->d_ctl.cap != ctl_cap_before) h->done_ready = false;  // fresh memory: the counter is garbage
    char* base = h->d_ctl.as<char>();
    if (reinterpret_cast<int32_t*>(base + status_bytes) != h->d_done) h->done_ready = false;  // W changed: it moved
    h->d_status = reinterpret_cast<int32_t*>(base);
    h->d_done = reinterpret_cast<int32_t*>(base + status_bytes);
    h->d_results = reinterpret_cast<LmResult*>(base + status_bytes + 16);
    if (h->h_ctl_cap < mirror_bytes) {
      if (h->h_ctl) (void)hipHostFree(h->h_ctl);
      h->h_ctl = nullptr;
      h->h_ctl_cap = 0;
      // coherent (fine-grained) whatever HIP_HOST_COHERENT says: the host polls the marks k_lm writes here
      // while the kernel may still be running (do_solve); non-coherent memory would only show them at its end
      HIP_TRY(hipHostMalloc((void**)&h->h_ctl, mirror_bytes + mirror_bytes / 4, hipHostMallocCoherent));
      h->h_ctl_cap = mirror_bytes + mirror_bytes / 4;
    }
    h->h_done = reinterpret_cast<int32_t*>(h->h_ctl);
    h->h_results = reinterpret_cast<LmResult*>(h->h_ctl + 16);
  }

  char* ha = h->h_arena;
  // One small window without a loss (small_window: the reference's shipped call): the kernel reads frames, starting
  // point and rays WHERE THE HOST WROTE THEM -- the arena is pinned and mapped into the device's address space, the
  // pairs' offsets travel with the kernel arguments (SolveParams::small_offs), so the launch's first loads are one
  // round trip over PCIe and there is no copy command in front of it.  With a loss the rays are read once per
  // evaluation: copied.  MVO_HIP_ARENA_COPY: always copied.
  // Only for the one-shot calls (mvo_hip_refine*: upload, ONE solve, download -- an upload that is solved many times
  // is better off in HBM) and only when every pair's rays are ONE load per lane of the wavefront that folds the pair
  // (<= 64 residual blocks: a second trip would be a second PCIe round trip; 900 blocks in three pairs measured 45.6 us
  // per solve in place against 41.3 us through the chip-wide fold, scripts/exp/inline_vs_fused.py).
  bool one_trip = true;
  for (const PairDev& pd : pairs) one_trip = one_trip && pd.count <= 64;
  const bool arena_in_place = h->one_shot && small_window(h) && one_trip && rays_in_arena && !rays_on_device &&
                              !no_rays && !h->env_arena_copy && arena_bytes <= (64u << 10);
  char* da = arena_in_place ? ha : h->d_arena.as<char>();
  // One-shot call, windows folded chip-wide, rays gathered into the arena (<= kArenaRayLimit blocks): the fold reads
  // every ray exactly once, so it reads them in the arena, over PCIe, and only the tables are copied (as the staged
  // path does for longer windows: zero_copy below).
  const bool rays_in_place = h->one_shot && !h->inline_eval && rays_in_arena && !rays_on_device && !no_rays &&
                             !arena_in_place && !h->env_arena_copy;
  std::memcpy(ha + o_frames, frames, sizeof(mvo_hip_frame) * (size_t)W * N);
  std::memcpy(ha + o_initx, initx.data(), sizeof(double) * initx.size());
  std::memcpy(ha + o_wins, wins.data(), sizeof(WinDev) * W);
  std::memcpy(ha + o_pairs, pairs.data(), sizeof(PairDev) * pairs.size());
  std::memcpy(ha + o_unit, unit.data(), sizeof(PairDev) * unit.size());
  if (!tiles.empty()) std::memcpy(ha + o_tiles, tiles.data(), sizeof(TileDev) * tiles.size());
  std::memcpy(ha + o_wave, wave_tiles.data(), sizeof(uint32_t) * wave_tiles.size());
  h->t.pair_u0 = reinterpret_cast<double*>(da + o_pu0);
  if (h->pass_mode)
    for (uint32_t w = 0; w < W; ++w) host_pair_u0(h, w, reinterpret_cast<double*>(ha + o_pu0) + 4 * (size_t)w * P);
  h->t.frames = reinterpret_cast<double*>(da + o_frames);
  h->t.initx = reinterpret_cast<double*>(da + o_initx);
  h->t.wins = reinterpret_cast<WinDev*>(da + o_wins);
  h->t.pairs = reinterpret_cast<PairDev*>(da + o_pairs);
  h->t.pairs_unit = reinterpret_cast<PairDev*>(da + o_unit);
  h->t.tiles = reinterpret_cast<TileDev*>(da + o_tiles);
  h->t.wave_tiles = reinterpret_cast<uint32_t*>(da + o_wave);
  h->t.m0 = rays_in_arena ? reinterpret_cast<double*>((rays_in_place ? ha : da) + o_m0) : h->d_m0.as<double>();
  h->t.m1 = rays_in_arena ? reinterpret_cast<double*>((rays_in_place ? ha : da) + o_m1) : h->d_m1.as<double>();
  // Rays in the handle's pinned staging block, too many for the arena: the fold reads them exactly once, so
  // it reads them where they are, over PCIe (pinned host memory is mapped into the device's address
  // space) -- no copy command at all; measured against hipMemcpyAsync from the same block in
  // profiles/r3/NOTES.md.  (MVO_HIP_STAGED_COPY keeps the copy, for that comparison.)
  const bool zero_copy = rays_pinned && !rays_in_arena && W == 1 && !std::getenv("MVO_HIP_STAGED_COPY");
  if (zero_copy) {
    h->t.m0 = const_cast<double*>(obs[0].m0);
    h->t.m1 = const_cast<double*>(obs[0].m1);
  }
  uint64_t pos = 0;
  for (uint32_t w = 0; w < W; ++w) {
    const uint64_t cnt = obs[w].pair_offsets[P];
    if (cnt && !rays_on_device && !zero_copy) {
      if (rays_in_arena) {
        std::memcpy(ha + o_m0 + sizeof(double) * 3 * pos, obs[w].m0, sizeof(double) * 3 * cnt);
        std::memcpy(ha + o_m1 + sizeof(double) * 3 * pos, obs[w].m1, sizeof(double) * 3 * cnt);
      } else {
        HIP_TRY(hipMemcpyAsync(h->t.m0 + 3 * pos, obs[w].m0, sizeof(double) * 3 * cnt, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(h->t.m1 + 3 * pos, obs[w].m1, sizeof(double) * 3 * cnt, hipMemcpyHostToDevice, s));
      }
    }
    pos += cnt;
  }
  if (!arena_in_place)
    HIP_TRY(hipMemcpyAsync(da, ha, (rays_on_device && rays_in_arena) || rays_in_place ? o_m0 : arena_bytes,
                           hipMemcpyHostToDevice, s));
  if (rays_in_arena || rays_on_device || rays_pinned) {
    h->copy_pending = true;  // pinned source: the copy is asynchronous and the solve is queued behind it
  } else {
    HIP_TRY(hipStreamSynchronize(s));  // the caller's ray arrays may change once this call returns
  }
  h->uploaded = true;
  h->timing.upload_seconds =
      std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  return MVO_HIP_OK;
}

uint32_t pass_blocks(const mvo_hip_context* h) { return (h->n_waves + FOLD_WAVES - 1) / FOLD_WAVES; }
size_t pass_lds(const mvo_hip_context* h) { return sizeof(double) * 8 * (size_t)h->max_block_tiles; }

// observation pass #0: c = (R0 m0) x (R1 m1); stores c (pass mode) or sums the moments (no loss).
// fused: the last workgroup leaves the per-pair sums in d_pairsum (k_lm then runs from_pair_sums), and
// in pass mode the fold is the evaluation pass at the starting point as well (pair_u0), its tails project
// every pair (d_proj).
hipError_t launch_reduce_slots(mvo_hip_context* h, const double* upairs = nullptr);
hipError_t launch_fold(mvo_hip_context* h, bool write_c, bool fused = false) {
  const uint32_t blocks = pass_blocks(h);
  if (!blocks) return hipSuccess;
  // fused: the per-pair sums are wanted in d_pairsum -- by the ticket tail of the kernel itself, or
  // (MVO_HIP_NO_TAILS, the tail's bit-for-bit reference in the tests) by a k_reduce_slots launch behind it
  const bool tails = fused && !h->env_no_tails;
  int32_t* ticket = tails ? h->d_pticket.as<int32_t>() : nullptr;
  double* pairsum = h->d_pairsum.as<double>();
  const bool pass0 = write_c && fused;
  const PassOut po{h->t.pair_u0, pass0 ? h->d_proj.as<double>() : nullptr, 0, 0};
#define MVO_FOLD_ARGS(cbuf, celems) \
  h->t.tiles, h->t.wave_tiles, h->n_waves, h->t.frames, h->t.m0, h->t.m1, cbuf, celems, h->d_partial.as<double>(), \
      ticket, h->t.pairs, pairsum, po
  if (pass0 && h->opt.use_loss_function)
    hipLaunchKernelGGL((k_fold<true, false, 2>), dim3(blocks), dim3(FOLD_THREADS), pass_lds(h), h->stream,
                       MVO_FOLD_ARGS(h->d_c.as<double>(), (size_t)h->c_elems));
  else if (pass0)
    hipLaunchKernelGGL((k_fold<true, false, 1>), dim3(blocks), dim3(FOLD_THREADS), pass_lds(h), h->stream,
                       MVO_FOLD_ARGS(h->d_c.as<double>(), (size_t)h->c_elems));
  else if (write_c)
    hipLaunchKernelGGL((k_fold<true, false, 0>), dim3(blocks), dim3(FOLD_THREADS), 0, h->stream,
                       MVO_FOLD_ARGS(h->d_c.as<double>(), (size_t)h->c_elems));
  else
    hipLaunchKernelGGL((k_fold<false, true, 0>), dim3(blocks), dim3(FOLD_THREADS), pass_lds(h), h->stream,
                       MVO_FOLD_ARGS((double*)nullptr, (size_t)0));
#undef MVO_FOLD_ARGS
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && fused && !tails) e = launch_reduce_slots(h, pass0 ? h->t.pair_u0 : nullptr);
  return e;
}

// one evaluation pass over the folded c at the unit baselines k_lm published (d_pair_u)
hipError_t launch_pass(mvo_hip_context* h, bool fused = false) {
  const uint32_t blocks = pass_blocks(h);
  if (!blocks) return hipSuccess;
  const bool tails = fused && !h->env_no_tails;
  int32_t* ticket = tails ? h->d_pticket.as<int32_t>() : nullptr;
  const PassOut po{h->d_pair_u.as<double>(), fused ? h->d_proj.as<double>() : nullptr, 0, 0};
  if (h->opt.use_loss_function)
    hipLaunchKernelGGL((k_pass<true>), dim3(blocks), dim3(FOLD_THREADS), pass_lds(h), h->stream,
                       h->t.tiles, h->t.wave_tiles, h->n_waves, h->d_c.as<double>(),
                       (size_t)h->c_elems, h->d_status,
                       h->d_partial.as<double>(), ticket, h->t.pairs, h->d_pairsum.as<double>(), po);
  else
    hipLaunchKernelGGL((k_pass<false>), dim3(blocks), dim3(FOLD_THREADS), pass_lds(h), h->stream,
                       h->t.tiles, h->t.wave_tiles, h->n_waves, h->d_c.as<double>(),
                       (size_t)h->c_elems, h->d_status,
                       h->d_partial.as<double>(), ticket, h->t.pairs, h->d_pairsum.as<double>(), po);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && fused && !tails) e = launch_reduce_slots(h, h->d_pair_u.as<double>());
  return e;
}

// which compile-time specialisation of k_lm serves these parameters (mvo_lm.h, "SPEC"); generic_only:
// MVO_HIP_LM_GENERIC at mvo_hip_create (tests compare the specialisations against the generic kernel)
static int lm_spec(const SolveParams& prm, bool generic_only) {
  if (generic_only || prm.cls_ready || prm.fix_length || prm.trace_cap != 0) return 0;
  if (prm.pass_mode && !prm.inline_eval && prm.proj_ready && prm.x0_ready) return 5;
  if (prm.use_loss || prm.pass_mode || prm.proj_ready) return 0;
  const int base = prm.inline_eval ? 2 : 1;
  return (prm.N - 1 - prm.M <= 32) ? base + 2 : base;
}
template <int NT>
static auto lm_kernel(int spec) -> decltype(&k_lm<NT, 0>) {
  if (NT == 64 && (spec == 3 || spec == 4)) spec -= 2;  // one wavefront: no class wavefronts to specialise
  return spec == 1 ? k_lm<NT, 1> : spec == 2 ? k_lm<NT, 2> : spec == 3 ? k_lm<NT, 3> : spec == 4 ? k_lm<NT, 4> :
         spec == 5 ? k_lm<NT, 5> : k_lm<NT, 0>;
}

template <int NT>
hipError_t launch_lm(mvo_hip_context* h, const SolveParams& prm, bool from_pair_sums = false) {
  hipLaunchKernelGGL(lm_kernel<NT>(lm_spec(prm, h->env_lm_generic)), dim3(h->W), dim3(NT), 0, h->stream, prm, from_pair_sums ? 1 : 0,
                     h->t.wins,
                     h->t.frames, h->t.initx,
                     from_pair_sums ? h->t.pairs_unit : h->t.pairs,
                     from_pair_sums ? h->d_pairsum.as<double>() : h->d_partial.as<double>(),
                     h->t.m0, h->t.m1, h->d_pairsum.as<double>(),
                     h->d_states.as<LmState>(), h->results_on_host ? h->h_results : h->d_results,
                     h->d_status, h->d_trace.as<TraceRec>(),
                     h->d_done, h->d_proj.as<double>(), h->d_cls.as<double>(), h->d_pair_u.as<double>(), RobustArgs{});
  return hipGetLastError();
}

// ---- the robust path of one window as one persistent launch (k_lm<256, 6>, mvo_lm.h) ----
// MVO_HIP_NO_PERSISTENT (read at mvo_hip_create) keeps the launch-per-evaluation path; so does a handle on which a
// persistent launch once ended without a result (persistent_failed).
bool robust_persistent_ok(const mvo_hip_context* h, const SolveParams& prm) {
  if (h->env_no_persistent || h->persistent_failed || h->env_lm_generic || h->env_no_tails) return false;
  if (h->W != 1 || !h->pass_mode || h->inline_eval || prm.fix_length || prm.trace_cap != 0 || prm.cls_ready) return false;
  const uint32_t blocks = (h->n_waves + FOLD_WAVES - 1) / FOLD_WAVES;
  return blocks >= 1 && blocks <= (uint32_t)kRobustMaxWorkgroups && h->P <= (uint32_t)kRobustMaxPairs &&
         h->max_block_tiles <= (uint32_t)kRobustMaxTiles && (int)blocks <= h->n_cus;
}
hipError_t launch_lm_persistent(mvo_hip_context* h, SolveParams prm) {
  const uint32_t blocks = (h->n_waves + FOLD_WAVES - 1) / FOLD_WAVES;
  hipError_t e = h->d_grid.reserve(256);
  if (e != hipSuccess) return e;
  // d_grid (64 words): [0] / [8] the counters of the persistent launches, used in turn -- every launch zeroes the other
  // one for its successor, so only the first launch on a handle (and the one after a failure) needs a clearing
  // command --, [32] their error word, [16] / [48] the fused launch's count and error word
  if (!h->fused_ready) {
    e = hipMemsetAsync(h->d_grid.p, 0, 256, h->stream);
    if (e != hipSuccess) return e;
    h->fused_count = 0;
    h->fused_ready = true;
    h->grid_seq = 0;
  }
  prm.fresh = 1;
  prm.win_base = 0;
  prm.x0_ready = 1;
  prm.proj_ready = 1;
  RobustArgs ra{};
  ra.tiles = h->t.tiles;
  ra.wave_tiles = h->t.wave_tiles;
  ra.cbuf = h->d_c.as<double>();
  ra.cstride = (size_t)h->c_elems;
  ra.partial = h->d_partial.as<double>();
  ra.pair_ticket = h->d_pticket.as<int32_t>();
  ra.pairs_slots = h->t.pairs;
  ra.pair_sums = h->d_pairsum.as<double>();
  ra.proj[0] = h->d_proj.as<double>();
  ra.proj[1] = h->d_proj.as<double>() + 10 * (size_t)h->W * h->P;
  ra.gs.count = h->d_grid.as<uint32_t>() + 8 * (h->grid_seq & 1u);
  ra.next_count = h->d_grid.as<uint32_t>() + 8 * ((h->grid_seq + 1) & 1u);
  ++h->grid_seq;
  ra.gs.error = h->d_grid.as<uint32_t>() + 32;
  ra.n_waves = h->n_waves;
  ra.n_wg = blocks;
  // test aid (MVO_HIP_PERSISTENT_TEST_FAIL, read at mvo_hip_create): a grid size the kernel refuses -- it returns at
  // once without a result, which is what the host sees when a wait has given up: the fall-back path under test
  if (h->env_persistent_test_fail) ra.n_wg = blocks + 1;
  ra.robust = h->opt.use_loss_function ? 1 : 0;
  hipLaunchKernelGGL((k_lm<MVO_LM_NT_LARGE, 6>), dim3(blocks), dim3(MVO_LM_NT_LARGE), 0, h->stream, prm, 1, h->t.wins,
                     h->t.frames, h->t.initx, h->t.pairs_unit, h->d_pairsum.as<double>(), h->t.m0, h->t.m1,
                     h->d_pairsum.as<double>(), h->d_states.as<LmState>(),
                     h->results_on_host ? h->h_results : h->d_results, h->d_status, h->d_trace.as<TraceRec>(), h->d_done,
                     h->d_proj.as<double>(), h->d_cls.as<double>(), h->d_pair_u.as<double>(), ra);
  h->lm_started = true;
  return hipGetLastError();
}

// ---- one small window without a loss: fold and LM loop as ONE launch (k_lm<256, 7>, mvo_lm.h) ----
// The grid is the fold's plus one: workgroup 0 sets the LM loop up and goes on with it once every pair is finished.  Only where the fold does
// not want more than one workgroup per CU anyway (k_lm's workgroups own their CU's registers) and the window takes
// the specialised kernel with the solver wavefront (lm_spec == 3).  Never while HIP events time the launches: the
// fold's own duration is what `roofline` is made of.
bool fused_ok(const mvo_hip_context* h, const SolveParams& prm) {
  if (h->env_no_fused || h->fused_failed || h->env_lm_generic || h->env_no_tails || h->profiling) return false;
  if (h->W != 1 || h->pass_mode || h->inline_eval || h->sums_external || prm.trace_cap != 0) return false;
  if (lm_spec(prm, false) != 3) return false;
  const uint32_t blocks = (h->n_waves + FOLD_WAVES - 1) / FOLD_WAVES;
  return blocks >= 1 && (int)blocks <= h->env_fused_max_blocks && h->max_block_tiles <= (uint32_t)kRobustMaxTiles &&
         h->h_pairs.size() == h->P;
}
hipError_t launch_lm_fused(mvo_hip_context* h, SolveParams prm) {
  const uint32_t blocks = (h->n_waves + FOLD_WAVES - 1) / FOLD_WAVES;
  hipError_t e = h->d_grid.reserve(256);
  if (e != hipSuccess) return e;
  if (!h->fused_ready) {
    e = hipMemsetAsync(h->d_grid.p, 0, 256, h->stream);
    if (e != hipSuccess) return e;
    h->fused_count = 0;
    h->fused_ready = true;
  }
  uint32_t nonempty = 0;
  for (const PairDev& pd : h->h_pairs) nonempty += pd.count != 0 ? 1u : 0u;
  prm.fresh = 1;
  prm.win_base = 0;
  RobustArgs ra{};
  ra.tiles = h->t.tiles;
  ra.wave_tiles = h->t.wave_tiles;
  ra.partial = h->d_partial.as<double>();
  ra.pair_ticket = h->d_pticket.as<int32_t>();
  ra.pairs_slots = h->t.pairs;
  ra.pair_sums = h->d_pairsum.as<double>();
  ra.gs.count = h->d_grid.as<uint32_t>() + 16;
  ra.gs.error = h->d_grid.as<uint32_t>() + 48;
  ra.n_waves = h->n_waves;
  ra.n_wg = blocks;
  ra.target = h->fused_count + nonempty;
  ra.spin_limit = kGridSpinLimit;
  // test aid (MVO_HIP_FUSED_TEST_FAIL, read at mvo_hip_create): a target the count cannot reach and a short wait --
  // workgroup 0 gives up without a result, which is what the fall-back path is for
  if (h->env_fused_test_fail) {
    ra.target += 1;
    ra.spin_limit = 2000;
  }
  h->fused_count += nonempty;
  hipLaunchKernelGGL((k_lm<MVO_LM_NT_LARGE, 7>), dim3(blocks + 1), dim3(MVO_LM_NT_LARGE), 0, h->stream, prm, 1, h->t.wins,
                     h->t.frames, h->t.initx, h->t.pairs_unit, h->d_pairsum.as<double>(), h->t.m0, h->t.m1,
                     h->d_pairsum.as<double>(), h->d_states.as<LmState>(),
                     h->results_on_host ? h->h_results : h->d_results, h->d_status, h->d_trace.as<TraceRec>(), h->d_done,
                     (const double*)nullptr, (const double*)nullptr, (double*)nullptr, ra);
  h->lm_started = true;
  return hipGetLastError();
}

// batched windows, no loss: fold workgroups [b0, b1) on the main stream
hipError_t launch_fold_range(mvo_hip_context* h, uint32_t b0, uint32_t b1) {
  const uint32_t w0 = b0 * FOLD_WAVES, w1 = std::min<uint32_t>(h->n_waves, b1 * FOLD_WAVES);
  if (w1 <= w0) return hipSuccess;
  hipLaunchKernelGGL((k_fold<false, true, 0>), dim3(b1 - b0), dim3(FOLD_THREADS), pass_lds(h), h->stream, h->t.tiles,
                     h->t.wave_tiles + w0, w1 - w0, h->t.frames, h->t.m0, h->t.m1, (double*)nullptr, (size_t)0,
                     h->d_partial.as<double>(), (int32_t*)nullptr, (const PairDev*)nullptr, (double*)nullptr, PassOut{});
  return hipGetLastError();
}

// ... and the whole LM loop of windows [win0, win1) on the second stream
template <int NT>
hipError_t launch_lm_range(mvo_hip_context* h, SolveParams prm, uint32_t win0, uint32_t win1, hipStream_t stream) {
  prm.fresh = 1;
  prm.win_base = (int32_t)win0;
  hipLaunchKernelGGL(lm_kernel<NT>(lm_spec(prm, h->env_lm_generic)), dim3(win1 - win0), dim3(NT), 0, stream, prm, 0, h->t.wins, h->t.frames,
                     h->t.initx, h->t.pairs, h->d_partial.as<double>(), h->t.m0, h->t.m1, h->d_pairsum.as<double>(),
                     h->d_states.as<LmState>(), h->d_results, h->d_status,
                     h->d_trace.as<TraceRec>(), h->d_done, (const double*)nullptr, (const double*)nullptr,
                     (double*)nullptr, RobustArgs{});
  return hipGetLastError();
}

// batches, no loss, at most 64 pairs per window, no trace: four windows per wavefront (mvo_lm4.h)
bool lm4_eligible(const mvo_hip_context* h) {
  if (h->pass_mode || h->inline_eval || h->P > 64 || h->W < 4 || h->trace_cap != 0) return false;
  if (h->opt.assembly_mode & MVO_HIP_ASSEMBLY_ONE_PER_WAVE) return false;
  if (h->opt.assembly_mode & MVO_HIP_ASSEMBLY_FOUR_PER_WAVE) return true;
  // one wavefront per SIMD (512 registers each): 1024 slots.  A k_lm4 wavefront runs ~1.1-1.45 x as long as a
  // k_lm<64> one (config-3 windows, eleven evaluations: 95-103 us against 66 us for a few hundred wavefronts, 88 us
  // for 1024 of them, scripts/exp/lm4_crossover.py) and carries four windows.
  // Up to 2048 windows k_lm4 carries TWO per wavefront (launch_lm4_range): one round of the slots either way, and the
  // shared step wins from ~640 windows on (config-3 windows, device time per batch, two per wavefront against one:
  // 512: 116 / 116 us, 768: 152 / 156, 1024: 175 / 189, 1536: 229 / 297, 2048: 278 / 357).
  if (h->W <= 2048) return h->W > 640;
  const double slots = 1024.0;
  const double rounds1 = std::ceil(h->W / slots), rounds4 = std::ceil(h->W / (4.0 * slots));
  return rounds1 > 1.45 * rounds4;
}
// Batches of windows with at most 64 pairs, folded chip-wide: the LM kernels add the slots of their pairs up themselves,
// the fold runs without ticket tails and is cut finer (mvo_hip_upload decides per upload: h->slots_plan).
// MVO_HIP_LM4_SUMS: never (round 3's plan: tails, per-pair sums).
bool slots_plan_static(const mvo_hip_context* h) {
  if (h->pass_mode || h->inline_eval || h->env_lm4_sums || h->P > h->env_slots_max_p) return false;
  return h->W > h->env_slots_min_w;
}
// marks_only: the launch covers every window of the solve and nothing else counts them: completion is checked on the
// records' marks (h->lm4_mark, new for every solve) instead of the done counter
hipError_t launch_lm4_range(mvo_hip_context* h, SolveParams prm, uint32_t win0, uint32_t win1, hipStream_t stream,
                            bool from_pair_sums, bool marks_only = false) {
  prm.fresh = 1;
  prm.win_base = (int32_t)win0;
  // windows per wavefront: four, or two when four would leave more than half of the 1024 SIMDs without a wavefront
  // (MVO_HIP_LM4_WPW = 2 | 4 forces either, for experiments)
  uint32_t wpw = win1 - win0 <= 2048 ? 2 : 4;
  if (const char* e = std::getenv("MVO_HIP_LM4_WPW")) wpw = std::atoi(e) == 2 ? 2 : 4;
  const uint32_t groups = (win1 - win0 + wpw - 1) / wpw;
  // the usual window (five refined frames, free lengths) has a kernel of its own; MVO_HIP_LM_GENERIC: never
  auto kern = prm.M == 5 && !prm.fix_length && !h->env_lm_generic ? k_lm4<5> : k_lm4<0>;
  hipLaunchKernelGGL(kern, dim3(groups), dim3(64), 0, stream, prm, (int)win1, from_pair_sums ? 1 : 0,
                     h->results_on_host ? 1 : 0, marks_only ? h->lm4_mark : kResultWritten, (int)wpw, h->t.frames,
                     h->t.initx, from_pair_sums ? h->t.pairs_unit : h->t.pairs,
                     from_pair_sums ? h->d_pairsum.as<double>() : h->d_partial.as<double>(),
                     h->results_on_host ? h->h_results : h->d_results, h->d_status,
                     marks_only ? (int32_t*)nullptr : h->d_done);
  return hipGetLastError();
}

hipError_t launch_lm_any(mvo_hip_context* h, SolveParams prm, bool from_pair_sums = false) {
  prm.fresh = h->lm_started ? 0 : 1;  // the first launch of a solve starts from a zeroed state
  h->lm_started = true;
  // Up to one window per CU the four-wavefront kernel is the faster one whatever the window's size: the
  // verdict on a point runs beside the block products, an inline pass takes a pair per wavefront.  Beyond
  // that a window with at most 64 pairs gets one wavefront (four windows per CU at a time).
  // (MVO_HIP_ASSEMBLY_ONE_PER_WAVE asks for the one-wavefront kernel by name: tests compare against it)
  const bool one_wave = h->P <= 64 && (h->W > 256 || (h->opt.assembly_mode & MVO_HIP_ASSEMBLY_ONE_PER_WAVE));
  return one_wave ? launch_lm<64>(h, prm, from_pair_sums) : launch_lm<MVO_LM_NT_LARGE>(h, prm, from_pair_sums);
}

// ---- observation-partitioned mode: fold / pass -> per-pair sums -> caller's
// all-reduce -> LM step from the summed moments.  Every call drains the stream before returning.
// upairs != nullptr (pass sums at those unit baselines): the pairs' projected moments as well (d_proj) -- the
// launch-of-its-own form of the passes' tails (MVO_HIP_NO_TAILS)
hipError_t launch_reduce_slots(mvo_hip_context* h, const double* upairs) {
  const uint32_t n = h->W * h->P;
  const PassOut po{upairs, upairs ? h->d_proj.as<double>() : nullptr, 0, 0};
  hipLaunchKernelGGL(k_reduce_slots, dim3((n * 8 + 255) / 256), dim3(256), 0, h->stream,
                     h->t.pairs, n, h->d_partial.as<double>(), h->d_pairsum.as<double>(), po);
  return hipGetLastError();
}

// partitioned pass mode: this rank's class totals (MVO_CLS_DOUBLES per window) from its pairs' projections
hipError_t launch_class_sums(mvo_hip_context* h) {
  hipLaunchKernelGGL(k_class_sums, dim3(h->W), dim3(64 * MVO_MAXM), 0, h->stream, (int)h->M, (int)h->N, (int)h->P,
                     h->d_proj.as<double>(), h->d_cls.as<double>());
  return hipGetLastError();
}

// Pass mode: what the ranks exchange is the CONTRACTED form -- the class totals of the projected
// moments, MVO_CLS_DOUBLES per window (k_class_sums), linear in each rank's share of the residual blocks --
// and k_lm starts at the prefix sums of the classes (SolveParams::cls_ready): no rank repeats the
// per-pair stage.  No loss: S_p never changes, ONE exchange of the 8 sums per pair per solve, then the
// whole LM loop in one launch (a contracted exchange would cost one collective per EVALUATION there --
// 10 more latencies of a collective to save a 3-us per-pair stage).
size_t partial_count(const mvo_hip_context* h) {
  return h->pass_mode ? (size_t)MVO_CLS_DOUBLES * h->W : 8 * (size_t)h->W * h->P;
}
// (in place: k_class_sums rewrites every entry of d_cls before each exchange; d_pairsum see pairsum_foreign)
void* partial_buffer(const mvo_hip_context* h) { return h->pass_mode ? h->d_cls.p : h->d_pairsum.p; }
int partial_reserve(mvo_hip_context* h) {
  if (h->pass_mode) HIP_TRY(h->d_cls.reserve(sizeof(double) * MVO_CLS_DOUBLES * h->W));
  return MVO_HIP_OK;
}
int clean_tickets(mvo_hip_context* h) {
  if (h->tickets_clean) return MVO_HIP_OK;
  // the whole blocks: later uploads may use more of them
  HIP_TRY(hipMemsetAsync(h->d_pticket.p, 0, h->d_pticket.cap, h->stream));
  h->tickets_clean = true;
  return MVO_HIP_OK;
}

int partial_begin(mvo_hip_context* h) {
  if (!h->uploaded) return MVO_HIP_ERR_STATE;
  h->results_on_host = false;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  const uint32_t n = h->W * h->P;
  if (h->trace_cap > 0) HIP_TRY(h->d_trace.reserve(sizeof(TraceRec) * (size_t)h->trace_cap));
  h->lm_started = false;
  // window status words and the done counter are adjacent: one fill
  HIP_TRY(hipMemsetAsync(h->d_status, 0, (size_t)((char*)h->d_done - (char*)h->d_status) + 16, s));
  h->done_base = 0;
  h->done_ready = false;  // until this partitioned solve has been seen to complete (it may be abandoned)
  h->solved = false;
  h->lm4_marked = false;
  {
    int rc = clean_tickets(h);
    if (rc == MVO_HIP_OK) rc = partial_reserve(h);
    if (rc != MVO_HIP_OK) return rc;
  }
  if (!h->pass_mode) {
    HIP_TRY(hipMemsetAsync(h->d_pairsum.p, 0, sizeof(double) * 8 * (size_t)n, s));
    HIP_TRY(launch_fold(h, false));
    HIP_TRY(launch_reduce_slots(h));
    h->pairsum_foreign = true;  // the caller's all-reduced sums come back into d_pairsum
  } else {
    // the fold is the pass at the starting point; its tails project this rank's share of every pair
    HIP_TRY(launch_fold(h, true, true));
    HIP_TRY(launch_class_sums(h));
  }
  HIP_TRY(hipStreamSynchronize(s));
  return MVO_HIP_OK;
}

int partial_pass(mvo_hip_context* h) {
  if (!h->uploaded || !h->pass_mode || !h->d_proj.p || !h->lm_started) return MVO_HIP_ERR_STATE;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  HIP_TRY(launch_pass(h, true));
  HIP_TRY(launch_class_sums(h));
  HIP_TRY(hipStreamSynchronize(s));
  return MVO_HIP_OK;
}

int partial_advance(mvo_hip_context* h, int32_t* done) {
  if (!h->uploaded || !partial_buffer(h)) return MVO_HIP_ERR_STATE;
  HIP_TRY(hipSetDevice(h->device));
  hipStream_t s = h->stream;
  SolveParams prm = make_params(h);
  prm.inline_eval = 0;
  prm.x0_ready = 1;  // pass mode: the fold evaluated the starting point
  prm.cls_ready = h->pass_mode ? 1 : 0;
  HIP_TRY(launch_lm_any(h, prm, true));
  HIP_TRY(hipMemcpyAsync(h->h_done, h->d_done, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const bool all = (uint32_t)h->h_done[0] - h->done_base >= h->W;
  if (all) {
    h->done_base = (uint32_t)h->h_done[0];
    HIP_TRY(hipMemcpyAsync(h->h_results, h->d_results, sizeof(LmResult) * h->W, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    h->done_ready = true;
    h->solved = true;
  }
  if (done) *done = all ? 1 : 0;
  return MVO_HIP_OK;
}

// The whole partitioned solve with the collective ON THE STREAM: fold / pass -> this rank's sums ->
// ncclAllReduce -> the identical LM step on every rank.  No host round trip per pass: the done counter is
// looked at every 12 evaluations, as in the single-GPU robust path.  Every rank issues the same sequence
// of collectives: the summed values are bitwise identical on all ranks, so every rank takes the same
// decisions and finishes at the same evaluation.  What travels: no loss -- 8 fp64 per pair, once per
// solve; pass mode -- MVO_CLS_DOUBLES fp64 per window and evaluation (see partial_count).
// FAIL-STOP: when a HIP or RCCL call fails in the middle of the sequence this rank returns an error and the
// OTHER ranks stay blocked in their next collective; the caller must tear the job down (mvo_hip_destroy /
// process exit), as after any failed collective.  N > 1 has only run in a world of one and as two handles on
// one GPU (tests/test_gpu_rccl.py, tests/test_gpu_partitioned.py): unverified on several GPUs.
int do_solve_partitioned(mvo_hip_context* h) {
  if (!h->uploaded || !h->comm) return MVO_HIP_ERR_STATE;
  h->results_on_host = false;
  const auto t_begin = std::chrono::steady_clock::now();
  HIP_TRY(hipSetDevice(h->device));
  const MvoRccl& rc = mvo_rccl();
  hipStream_t s = h->stream;
  if (h->trace_cap > 0) HIP_TRY(h->d_trace.reserve(sizeof(TraceRec) * (size_t)h->trace_cap));
  h->lm_started = false;
  h->solved = false;
  h->lm4_marked = false;
  HIP_TRY(hipMemsetAsync(h->d_status, 0, (size_t)((char*)h->d_done - (char*)h->d_status) + 16, s));
  h->done_base = 0;
  h->done_ready = false;
  {
    int rc = clean_tickets(h);
    if (rc == MVO_HIP_OK) rc = partial_reserve(h);
    if (rc != MVO_HIP_OK) return rc;
  }
  SolveParams prm = make_params(h);
  prm.inline_eval = 0;  // the passes are launches of their own: their sums are what gets all-reduced
  prm.x0_ready = 1;
  prm.cls_ready = h->pass_mode ? 1 : 0;
  auto allreduce = [&]() -> int {
    void* buf = partial_buffer(h);
    const ncclResult_t r = rc.AllReduce(buf, buf, partial_count(h), ncclDouble, ncclSum, h->comm, s);
    if (r != ncclSuccess) {
      h->last_rccl_error = r;
      h->comm_broken = true;  // the peers are waiting in this collective: mvo_hip_comm_destroy aborts instead of draining
      return MVO_HIP_ERR_DEVICE;
    }
    return MVO_HIP_OK;
  };
  int lm_launches = 0, asm_launches = 0, collectives = 0;
  if (!h->pass_mode) {
    // S_p never changes without a loss: ONE collective per solve, then the whole LM loop in one launch
    HIP_TRY(launch_fold(h, false));
    ++asm_launches;
    HIP_TRY(launch_reduce_slots(h));
    h->pairsum_foreign = true;
    int rcode = allreduce();
    if (rcode != MVO_HIP_OK) return rcode;
    ++collectives;
    HIP_TRY(launch_lm_any(h, prm, true));
    ++lm_launches;
  } else {
    // evaluation 0 is the fold (the pass at the starting point); then one pass per candidate
    const int total_evals = h->opt.max_num_iterations + 1;
    const int chunk = 12;
    int issued = 0;
    while (issued < total_evals) {
      const int m = (total_evals - issued) < chunk ? (total_evals - issued) : chunk;
      for (int i = 0; i < m; ++i) {
        if (issued + i == 0) HIP_TRY(launch_fold(h, true, true));
        else HIP_TRY(launch_pass(h, true));
        ++asm_launches;
        HIP_TRY(launch_class_sums(h));
        int rcode = allreduce();
        if (rcode != MVO_HIP_OK) return rcode;
        ++collectives;
        HIP_TRY(launch_lm_any(h, prm, true));
        ++lm_launches;
      }
      issued += m;
      if (issued < total_evals) {
        HIP_TRY(hipMemcpyAsync(h->h_done, h->d_done, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if ((uint32_t)h->h_done[0] >= h->W) break;  // identical on every rank: the sums were
      }
    }
  }
  HIP_TRY(hipMemcpyAsync(h->h_ctl, h->d_done, 16 + sizeof(LmResult) * h->W, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  h->copy_pending = false;
  if ((uint32_t)h->h_done[0] != h->W) return MVO_HIP_ERR_DEVICE;
  h->done_base = (uint32_t)h->h_done[0];
  h->done_ready = true;
  h->solved = true;
  mvo_hip_timing& tm = h->timing;
  tm.solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  tm.device_seconds = tm.assembly_seconds = tm.assembly_first_seconds = 0.0;
  tm.assembly_launches = asm_launches;
  tm.lm_launches = lm_launches;
  tm.residual_blocks = h->n_obs;
  tm.assembly_bytes = 0;
  (void)collectives;
  return MVO_HIP_OK;
}

int do_solve(mvo_hip_context* h);
// A persistent launch that ended without a result (a wait gave up -- some workgroup of the grid was not resident,
// e.g. another process holds CUs): the stream is idle here; the handle stops trying and the solve is repeated
// launch by launch.  do_solve treats the half-counted tickets and status words like those of any abandoned solve.
static int persistent_fallback(mvo_hip_context* h) {
  h->persistent_failed = true;
  h->persistent_in_flight = false;
  h->fused_ready = false;  // (d_grid: a counter half counted, the error word raised)
  h->done_ready = false;
  h->tickets_clean = false;
  return do_solve(h);
}

// the fused launch ended without a result (its workgroup 0 gave up waiting): two launches from now on
static int fused_fallback(mvo_hip_context* h) {
  h->fused_failed = true;
  h->fused_in_flight = false;
  h->fused_ready = false;
  h->done_ready = false;
  h->tickets_clean = false;
  return do_solve(h);
}

int do_solve(mvo_hip_context* h) {
  if (!h->uploaded) return MVO_HIP_ERR_STATE;
  const auto t_begin = std::chrono::steady_clock::now();
  HIP_TRY(hipSetDevice(h->device));
  const bool prof = h->profiling;
  if (prof) {
    int rc = ensure_events(h);
    if (rc != MVO_HIP_OK) return rc;
  }
  hipStream_t s = h->stream;
  const SolveParams prm = make_params(h);
  if (h->trace_cap > 0) HIP_TRY(h->d_trace.reserve(sizeof(TraceRec) * (size_t)h->trace_cap));

  h->lm_started = false;
  h->persistent_in_flight = false;
  h->fused_in_flight = false;
  // No per-solve fills: the first k_lm launch starts from a zero state of its own and clears its
  // window's status word (SolveParams::fresh); the done counter only ever counts up (done_base).
  if (!h->done_ready) {
    HIP_TRY(hipMemsetAsync(h->d_status, 0, (size_t)((char*)h->d_done - (char*)h->d_status) + 16, s));
    h->done_base = 0;
    h->tickets_clean = false;  // an abandoned solve may have left a pass half counted
  }
  {
    int rc = clean_tickets(h);
    if (rc != MVO_HIP_OK) return rc;
  }
  if (h->pairsum_foreign) {
    // a partitioned no-loss solve all-reduced into d_pairsum: the rows of pairs that are empty on this rank
    // hold the other ranks' sums, and no fold rewrites a pair without slots
    HIP_TRY(hipMemsetAsync(h->d_pairsum.p, 0, sizeof(double) * 8 * (size_t)h->W * h->P, s));
    h->pairsum_foreign = false;
  }
  h->done_ready = false;  // until this solve has been seen to complete
  h->solved = false;
  h->lm4_marked = false;
  const uint32_t done_base = h->done_base;
  // Up to 1024 windows (MVO_HIP_HOST_RESULTS_MAX_W) the result records (200 B each) are written by the LM kernel
  // straight into the pinned host mirror: no copy command and no wait for the stream behind the kernel (~4 us of a
  // 45-us default-size solve; config-3 windows, call time: 128 / 256 / 512 windows 75.0 -> 63.0, 87.8 -> 77.3, 120.9 ->
  // 113.8 us, 704 / 768 / 896 / 1024 windows 153.1 -> 139.8, 158.2 -> 147.1, 169.8 -> 162.4, 181.8 -> 177.6; the limit
  // was 64 until the end of round 3).  Beyond, k_lm4's marks-only epilogue and ONE copy are the faster (1536 windows:
  // 229.9 us copied against 232.3, 2048: 277.4 against 288.0, 4096: 502 against 524).
  // (the chunked experiment paths -- MVO_HIP_OVERLAP, MVO_HIP_PIPELINE -- count finished windows and copy the records)
  // (under the loss -- a launch per evaluation -- the same: 128 / 512 / 1024 windows of 10 500 blocks 289.8 -> 267.5,
  // 533.2 -> 516.2, 896.1 -> 885.6 us per call, scripts/exp/batch_inline_vs_tiled_loss.py; a solve longer than the
  // 0.5 ms the host polls for ends in a wait for the stream as before, still without the copy command)
  h->results_on_host = h->W <= (h->pass_mode ? h->env_host_results_max_w_loss : h->env_host_results_max_w) &&
                       !h->env_no_host_results &&  // MVO_HIP_NO_HOST_RESULTS
                       !(h->n_chunks > 1 && !h->pass_mode && !h->inline_eval);
  if (h->results_on_host)
    for (uint32_t w = 0; w < h->W; ++w) h->h_results[w].pad = 0;

  const uint32_t fold_blocks = pass_blocks(h);
  int n_timed = 0;
  auto timed_begin = [&]() -> hipError_t {
    return prof && n_timed < kMaxTimedLaunches ? hipEventRecord(h->ev[2 + 2 * n_timed], s) : hipSuccess;
  };
  auto timed_end = [&]() -> hipError_t {
    hipError_t e = prof && n_timed < kMaxTimedLaunches ? hipEventRecord(h->ev[3 + 2 * n_timed], s) : hipSuccess;
    ++n_timed;
    return e;
  };

  if (prof) HIP_TRY(hipEventRecord(h->ev[0], s));
  int lm_launches = 0, asm_launches = 0;
  if (h->inline_eval) {
    // the whole solve -- observation passes included -- is one launch, one workgroup per window
    HIP_TRY(launch_lm_any(h, prm, true));
    ++lm_launches;
  } else if (!h->pass_mode && h->n_chunks > 1) {
    // many windows: fold and LM loops overlap chunk by chunk on two streams
    if (!h->overlap_ready) {
      for (auto& q : h->stream_lm) HIP_TRY(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
      for (auto& e : h->ev_chunk) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      h->overlap_ready = true;
    }
    uint32_t b0 = 0, win0 = 0;
    bool used[3] = {false, false, false};
    for (int c = 0; c < h->n_chunks; ++c) {
      const uint32_t b1 = h->chunk_block_end[c], win1 = h->chunk_win_end[c];
      HIP_TRY(timed_begin());
      HIP_TRY(launch_fold_range(h, b0, b1));
      HIP_TRY(timed_end());
      ++asm_launches;
      if (win1 > win0) {
        // the staggered plan's last chunk runs behind the fold on the fold's own stream; the others beside it
        const bool last_inline = h->pipelined && c + 1 == h->n_chunks;
        const int qi = h->pipelined ? c % 3 : c & 1;
        hipStream_t q = last_inline ? s : h->stream_lm[qi];
        if (!last_inline) {
          used[qi] = true;
          HIP_TRY(hipEventRecord(h->ev_chunk[c], s));
          HIP_TRY(hipStreamWaitEvent(q, h->ev_chunk[c], 0));
        }
        int kind = h->pipelined ? h->chunk_kind[c] : (lm4_eligible(h) ? 4 : (h->P <= 64 ? 64 : 256));
        if (kind == 4 && h->trace_cap != 0) kind = 64;  // k_lm4 records no trace (asked for after the upload)
        if (kind == 4) HIP_TRY(launch_lm4_range(h, prm, win0, win1, q, false));
        else if (kind == 64) HIP_TRY(launch_lm_range<64>(h, prm, win0, win1, q));
        else HIP_TRY(launch_lm_range<MVO_LM_NT_LARGE>(h, prm, win0, win1, q));
        ++lm_launches;
      }
      b0 = b1;
      win0 = win1;
    }
    h->lm_started = true;
    for (int q = 0; q < 3; ++q)
      if (used[q]) {
        HIP_TRY(hipEventRecord(h->ev_chunk[kMaxOverlapChunks + q], h->stream_lm[q]));
        HIP_TRY(hipStreamWaitEvent(s, h->ev_chunk[kMaxOverlapChunks + q], 0));
      }
  } else if (!h->pass_mode) {
    // one fold+moment pass, then the whole LM loop in a single launch.  S_p never changes without a
    // loss: the fold's last workgroup adds the slots up per pair, every k_lm lane loads the sums of its
    // (up to four) pairs once and keeps them in registers.
    h->fused_in_flight = fold_blocks != 0 && fused_ok(h, prm);
    if (h->fused_in_flight) {
      HIP_TRY(launch_lm_fused(h, prm));  // fold and LM loop in one launch
    } else if (fold_blocks && !h->sums_external) {
      HIP_TRY(timed_begin());
      // (k_lm4 adds the slots of its pairs up itself: no ticket tails, see the tiling in mvo_hip_upload)
      HIP_TRY(launch_fold(h, false, !(h->slots_plan && !h->sums_external)));
      HIP_TRY(timed_end());
      ++asm_launches;
    }
    if (h->fused_in_flight) {
    } else if (lm4_eligible(h) && fold_blocks) {
      // (records polled in the host mirror carry the fixed mark: the host clears them before every solve)
      h->lm4_marked = !h->results_on_host;
      if (h->lm4_marked) h->lm4_mark = kResultWritten + 0x10000 * (int32_t)(1 + (h->lm4_seq++ & 0xfff));
      HIP_TRY(launch_lm4_range(h, prm, 0, h->W, s, !h->slots_plan || h->sums_external, h->lm4_marked));
      h->lm_started = true;
    } else {
      HIP_TRY(launch_lm_any(h, prm, (fold_blocks != 0 && !h->slots_plan) || h->sums_external));
    }
    ++lm_launches;
  } else {
    // robust / pass mode: max_it + 1 evaluations (x0 and one candidate per iteration), each one pass over
    // c and one k_lm launch that reads the pass's per-pair sums.  The fold is the pass at x0.
    SolveParams pp = prm;
    const bool fused = fold_blocks != 0;
    if (fused) {
      HIP_TRY(timed_begin());
      HIP_TRY(launch_fold(h, true, true));
      HIP_TRY(timed_end());
      ++asm_launches;
    }
    // The fold is the pass at the starting point; fold and passes leave every pair projected in d_proj (their
    // tails) and k_lm the unit baselines of the next point in d_pair_u.  Without any residual block there
    // is no pass at all: the projections are the zeros the upload left.
    pp.x0_ready = 1;
    pp.proj_ready = 1;
    h->persistent_in_flight = fused && robust_persistent_ok(h, pp);
    if (h->persistent_in_flight) {
      // one window whose pass grid fits the chip in one round: the whole solve behind the fold is ONE launch, every
      // workgroup of the pass grid running the LM step itself (k_lm<256, 6>, mvo_lm.h)
      HIP_TRY(launch_lm_persistent(h, pp));
    } else {
      HIP_TRY(launch_lm_any(h, pp, true));  // evaluates x0 and takes the first step
    }
    ++lm_launches;
    // enqueue the evaluations in chunks without any host round-trip inside a chunk, and look at the done
    // counter between
    const int total_evals = h->persistent_in_flight ? 0 : h->opt.max_num_iterations;
    const int chunk = 12;
    int issued = 0;
    while (issued < total_evals) {
      const int n = (total_evals - issued) < chunk ? (total_evals - issued) : chunk;
      for (int i = 0; i < n; ++i) {
        if (fold_blocks) {
          HIP_TRY(timed_begin());
          HIP_TRY(launch_pass(h, true));
          HIP_TRY(timed_end());
          ++asm_launches;
        }
        HIP_TRY(launch_lm_any(h, pp, true));
        ++lm_launches;
      }
      issued += n;
      if (issued < total_evals) {
        HIP_TRY(hipMemcpyAsync(h->h_done, h->d_done, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if ((uint32_t)h->h_done[0] - done_base >= h->W) break;
      }
    }
  }
  if (prof) HIP_TRY(hipEventRecord(h->ev[1], s));
  if (h->results_on_host) {
    // a few windows: k_lm wrote the result records into the pinned mirror itself (and marked each: pad,
    // behind a system-scope release); no copy command, the done counter is accounted for without being
    // read.  The host polls the marks for up to half a millisecond before it falls back on waiting for
    // the stream (config 2: 75.9 -> 74.2 us per solve).  What is still queued behind the marks (the
    // kernel's end) is ordered before anything the next call queues on the same stream.
    bool seen = false;
    if (!prof && !h->env_no_poll) {  // MVO_HIP_NO_POLL
      const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(500);
      for (uint32_t spins = 0; !seen; ++spins) {
        seen = true;
        for (uint32_t w = 0; w < h->W && seen; ++w)
          seen = __atomic_load_n(&h->h_results[w].pad, __ATOMIC_ACQUIRE) == kResultWritten;
        if (seen) break;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((spins & 63u) == 63u && std::chrono::steady_clock::now() > deadline) break;
      }
    }
    if (!seen) HIP_TRY(hipStreamSynchronize(s));
    h->copy_pending = false;
    for (uint32_t w = 0; w < h->W; ++w)
      if (h->h_results[w].pad != kResultWritten) {
        if (h->persistent_in_flight) return persistent_fallback(h);
        if (h->fused_in_flight) return fused_fallback(h);
        return MVO_HIP_ERR_DEVICE;  // every window must have terminated
      }
    h->h_done[0] = (int32_t)(done_base + h->W);
  } else {
    // done counter and results are adjacent on both sides: one copy
    HIP_TRY(hipMemcpyAsync(h->h_ctl, h->d_done, 16 + sizeof(LmResult) * h->W, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    h->copy_pending = false;
    if (h->lm4_marked) {
      // k_lm4 over the whole batch counts nothing (launch_lm4_range): every record must carry this solve's mark
      for (uint32_t w = 0; w < h->W; ++w) {
        if (w + 16 < h->W) __builtin_prefetch(&h->h_results[w + 16].pad);  // (freshly copied: every record is a miss)
        if (h->h_results[w].pad != h->lm4_mark) return MVO_HIP_ERR_DEVICE;
      }
      h->h_done[0] = (int32_t)done_base;  // (the counter did not move)
    } else if ((uint32_t)h->h_done[0] - done_base != h->W) {
      if (h->persistent_in_flight) return persistent_fallback(h);
      if (h->fused_in_flight) return fused_fallback(h);
      return MVO_HIP_ERR_DEVICE;  // every window must have terminated
    }
  }
  h->done_base = (uint32_t)h->h_done[0];
  h->done_ready = true;
  h->solved = true;

  mvo_hip_timing& tm = h->timing;
  tm.solve_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
  tm.device_seconds = 0.0;
  tm.assembly_seconds = 0.0;
  tm.assembly_first_seconds = 0.0;
  tm.assembly_launches = asm_launches;
  tm.lm_launches = lm_launches;
  tm.residual_blocks = h->n_obs;
  tm.assembly_bytes = 0;
  if (!prof) return MVO_HIP_OK;
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, h->ev[0], h->ev[1]));
  tm.device_seconds = ms * 1e-3;
  const int nt = n_timed < kMaxTimedLaunches ? n_timed : kMaxTimedLaunches;
  double pass_seconds = 0.0;
  for (int i = 0; i < nt; ++i) {
    HIP_TRY(hipEventElapsedTime(&ms, h->ev[2 + 2 * i], h->ev[3 + 2 * i]));
    if (i == 0) tm.assembly_first_seconds = ms * 1e-3;
    else pass_seconds += ms * 1e-3;
  }
  if (h->persistent_in_flight) {
    // the passes run inside the persistent launch (k_lm<256, 6>): one pass per LM iteration; the launch -- everything
    // behind the fold -- is attributed to them (an upper bound on their time: steps and hand-overs are in it)
    tm.assembly_seconds = tm.device_seconds - tm.assembly_first_seconds;
    tm.assembly_launches = 1;
    tm.assembly_bytes = 24ull * h->n_obs * (uint64_t)(h->h_results[0].iterations > 0 ? h->h_results[0].iterations : 0);
  } else if (h->inline_eval) {
    // the passes run inside k_lm: attribute the launch to them (an upper bound on their time)
    tm.assembly_seconds = tm.device_seconds;
    tm.assembly_launches = 1;
    uint64_t evals = 0;
    for (uint32_t w = 0; w < h->W; ++w) evals += (uint64_t)(h->h_results[w].iterations + 1);
    tm.assembly_bytes = h->pass_mode ? (uint64_t)(48.0 * (double)h->n_obs * (double)evals / (double)h->W)
                                     : 48ull * h->n_obs;
  } else if (!h->pass_mode && h->n_chunks > 1) {
    // the fold ran as n_chunks launches next to the LM loops of the chunks before: average launch
    tm.assembly_seconds = tm.assembly_first_seconds + pass_seconds;
    tm.assembly_launches = nt;
    tm.assembly_bytes = 48ull * h->n_obs / (uint64_t)h->n_chunks;
  } else if (!h->pass_mode) {
    tm.assembly_seconds = tm.assembly_first_seconds;
    tm.assembly_bytes = 48ull * h->n_obs;  // k_fold<moments>: read m0, m1
  } else {
    tm.assembly_seconds = pass_seconds;    // k_pass launches only (the fold is reported apart)
    tm.assembly_launches = nt > 0 ? nt - 1 : 0;
    tm.assembly_bytes = 24ull * h->n_obs;  // k_pass: read c
  }
  return MVO_HIP_OK;
}

int do_download(mvo_hip_context* h, mvo_hip_frame* frames, mvo_hip_summary* summaries) {
  if (!h->uploaded || !h->solved) return MVO_HIP_ERR_STATE;  // nothing solved yet: the result records are stale
  for (uint32_t w = 0; w < h->W; ++w) {
    const LmResult& r = h->h_results[w];
    // (the records were just written by the copy engine or the GPU: every line is a miss; ask for them ahead)
    if (w + 8 < h->W) {
      const char* nx = reinterpret_cast<const char*>(&h->h_results[w + 8]);
      __builtin_prefetch(nx);
      __builtin_prefetch(nx + 64);
      __builtin_prefetch(nx + 128);
      __builtin_prefetch(nx + 192);
      if (frames)  // ... and the caller's frame records, which are written in part (lines to be owned first)
        for (uint32_t k = 0; k < h->M; ++k) __builtin_prefetch(&frames[(size_t)(w + 8) * h->N + k].vec[0], 1);
    }
    if (frames) {
      // IR.cpp:240-246: refined frames get scale = s(t) and the optimised direction
      for (uint32_t k = 0; k < h->M; ++k) {
        mvo_hip_frame& f = frames[(size_t)w * h->N + k];
        f.scale = r.scale[k];  // s(t) of scaleTemplated, evaluated by the kernel that terminated the window
        f.vec[0] = r.vec[k][0];
        f.vec[1] = r.vec[k][1];
        f.vec[2] = r.vec[k][2];
      }
    }
    if (summaries) {
      mvo_hip_summary& sm = summaries[w];
      sm.termination = r.termination;
      sm.reason = r.reason;
      sm.iterations = r.iterations;
      sm.num_successful_steps = r.nsucc;
      sm.num_unsuccessful_steps = r.nunsucc;
      sm.clamped_scales = h->v_clamped[w];
      sm.initial_cost = r.initial_cost;
      sm.final_cost = r.final_cost;
    }
  }
  return MVO_HIP_OK;
}

}  // namespace

// =========================================================================================
// C ABI
// =========================================================================================
// mvo_hip_debug_poison_lds: every workgroup fills its LDS with quiet NaNs and lingers a little so that
// the launch spreads over all CUs instead of recycling the first few
__global__ void k_poison_lds(int n_doubles) {
  extern __shared__ double poison[];
  for (int i = threadIdx.x; i < n_doubles; i += blockDim.x) poison[i] = __longlong_as_double(0x7ff8dead0000beefLL);
  __syncthreads();
  if (poison[(threadIdx.x * 37) % n_doubles] == 0.0) poison[0] = 1.0;  // keeps the stores alive
  __builtin_amdgcn_s_sleep(64);
}

extern "C" {

int mvo_hip_create(int device_ordinal, uint32_t max_windows, uint32_t max_frames, uint64_t max_observations,
                   mvo_hip_handle* out) {
  if (!out) return MVO_HIP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MVO_HIP_ERR_NO_DEVICE;
  if (device_ordinal < 0 || device_ordinal >= count) return MVO_HIP_ERR_INVALID_ARGUMENT;
  mvo_hip_context* h = new (std::nothrow) mvo_hip_context();
  if (!h) return MVO_HIP_ERR_OUT_OF_MEMORY;
  h->device = device_ordinal;
  if (const char* e = std::getenv("MVO_HIP_WAVES_PER_CU")) {
    const int t = std::atoi(e);
    if (t >= 1 && t <= 1024) h->env_waves_per_cu = (uint32_t)t;
  }
  if (const char* e = std::getenv("MVO_HIP_OVERLAP_CHUNKS")) {
    const int t = std::atoi(e);
    if (t >= 2 && t <= kMaxOverlapChunks) h->env_overlap_chunks = t;
  }
  h->env_no_host_results = std::getenv("MVO_HIP_NO_HOST_RESULTS") != nullptr;
  if (const char* e = std::getenv("MVO_HIP_HOST_RESULTS_MAX_W"))
    h->env_host_results_max_w = h->env_host_results_max_w_loss = (uint32_t)std::atoi(e);
  h->env_no_poll = std::getenv("MVO_HIP_NO_POLL") != nullptr;
  h->env_no_tails = std::getenv("MVO_HIP_NO_TAILS") != nullptr;
  h->env_arena_copy = std::getenv("MVO_HIP_ARENA_COPY") != nullptr;

  h->env_lm4_sums = std::getenv("MVO_HIP_LM4_SUMS") != nullptr;
  if (const char* e = std::getenv("MVO_HIP_SLOTS_MIN_W")) h->env_slots_min_w = (uint32_t)std::atoi(e);
  if (const char* e = std::getenv("MVO_HIP_SLOTS_MAX_P")) h->env_slots_max_p = (uint32_t)std::atoi(e);
  h->env_lm_generic = std::getenv("MVO_HIP_LM_GENERIC") != nullptr;
  h->env_no_persistent = std::getenv("MVO_HIP_NO_PERSISTENT") != nullptr;
  h->env_no_fused = std::getenv("MVO_HIP_NO_FUSED") != nullptr;
  h->env_fused_test_fail = std::getenv("MVO_HIP_FUSED_TEST_FAIL") != nullptr;
  h->env_persistent_test_fail = std::getenv("MVO_HIP_PERSISTENT_TEST_FAIL") != nullptr;
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess) h->n_cus = cus;
  }
  // fold and LM loop as one launch up to one fold workgroup per CU (scripts/exp/fused_limit.py, config 2's shape, one
  // launch / two: 120 k blocks 51.0 / 51.6 us, 168 k 50.7 / 51.6, 239 k 51.1 / 52.0, 359 k 55.5 / 53.2, 726 k 64.3 / 57.8)
  h->env_fused_max_blocks = h->n_cus;
  if (const char* e = std::getenv("MVO_HIP_FUSED_MAX_BLOCKS")) h->env_fused_max_blocks = std::atoi(e);
  if (hipSetDevice(device_ordinal) != hipSuccess ||
      hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
      mvo_noalloc_resolve(device_ordinal) != hipSuccess) {  // the library's set-up call: scratch address, once per device
    delete h;
    return MVO_HIP_ERR_DEVICE;
  }
  // capacity hints: pre-size the large buffers so that the first refine() allocates nothing big
  if (max_observations) {
    (void)h->d_m0.reserve(sizeof(double) * 3 * (max_observations + 1));
    (void)h->d_m1.reserve(sizeof(double) * 3 * (max_observations + 1));
  }
  if (max_windows && max_frames) (void)h->d_arena.reserve(sizeof(mvo_hip_frame) * (size_t)max_windows * max_frames);
  *out = h;
  return MVO_HIP_OK;
}

void mvo_hip_destroy(mvo_hip_handle h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->comm) (void)mvo_rccl().CommDestroy(h->comm);
  h->comm = nullptr;
  DevBuf* bufs[] = {&h->d_arena, &h->d_m0, &h->d_m1, &h->d_c, &h->d_partial, &h->d_pair_u, &h->d_states,
                    &h->d_ctl, &h->d_trace, &h->d_res, &h->d_jac, &h->d_pairsum, &h->d_pticket, &h->d_proj, &h->d_cls, &h->d_grid};
  for (DevBuf* b : bufs) b->release();
  if (h->h_arena) (void)hipHostFree(h->h_arena);
  if (h->h_ctl) (void)hipHostFree(h->h_ctl);
  if (h->h_st_offs) (void)hipHostFree(h->h_st_offs);
  if (h->h_st_m0) (void)hipHostFree(h->h_st_m0);
  if (h->h_st_m1) (void)hipHostFree(h->h_st_m1);
  if (h->events_ready)
    for (auto& e : h->ev) (void)hipEventDestroy(e);
  if (h->overlap_ready) {
    for (auto& e : h->ev_chunk) (void)hipEventDestroy(e);
    for (auto& q : h->stream_lm) (void)hipStreamDestroy(q);
  }
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int mvo_hip_upload(mvo_hip_handle h, uint32_t n_windows, const mvo_hip_frame* frames, uint32_t M, uint32_t N,
                   const mvo_hip_options* options, const mvo_hip_observations* observations) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  return do_upload(h, n_windows, frames, M, N, options, observations);
}

// library-internal (csrc/mvo_internal.h): the device-side gather queues its kernels on the solver's
// stream and hands over device-resident rays
void* mvo_hip_internal_stream(mvo_hip_handle h) { return h ? (void*)h->stream : nullptr; }
int mvo_hip_internal_device(mvo_hip_handle h) { return h ? h->device : -1; }
int mvo_hip_internal_upload_device_rays(mvo_hip_handle h, const mvo_hip_frame* frames, uint32_t M, uint32_t N,
                                        const mvo_hip_options* options, const mvo_hip_observations* observations,
                                        double** m0_device, double** m1_device) {
  if (!h || !m0_device || !m1_device) return MVO_HIP_ERR_INVALID_ARGUMENT;
  const int rc = do_upload(h, 1, frames, M, N, options, observations, true);
  if (rc != MVO_HIP_OK) return rc;
  *m0_device = h->t.m0;
  *m1_device = h->t.m1;
  return MVO_HIP_OK;
}

int mvo_hip_internal_upload_pair_sums(mvo_hip_handle h, const mvo_hip_frame* frames, uint32_t M, uint32_t N,
                                      const mvo_hip_options* options, const mvo_hip_observations* observations,
                                      const double** frames_device, double** pair_sums_device) {
  if (!h || !options || !frames_device || !pair_sums_device) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (options->use_loss_function || (options->assembly_mode & MVO_HIP_ASSEMBLY_PASSES)) return MVO_HIP_ERR_UNSUPPORTED;
  mvo_hip_options opt = *options;
  opt.assembly_mode |= MVO_HIP_ASSEMBLY_TILED;  // the window's own workgroup must not go looking for rays
  const int rc = do_upload(h, 1, frames, M, N, &opt, observations, true, true);
  if (rc != MVO_HIP_OK) return rc;
  h->sums_external = true;
  *frames_device = h->t.frames;
  *pair_sums_device = h->d_pairsum.as<double>();
  return MVO_HIP_OK;
}

int mvo_hip_solve(mvo_hip_handle h) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  return do_solve(h);
}

int mvo_hip_download(mvo_hip_handle h, mvo_hip_frame* frames, mvo_hip_summary* summaries) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  return do_download(h, frames, summaries);
}

int mvo_hip_refine_batch(mvo_hip_handle h, uint32_t n_windows, mvo_hip_frame* frames, uint32_t M, uint32_t N,
                         const mvo_hip_options* options, const mvo_hip_observations* observations,
                         mvo_hip_summary* summaries) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  h->one_shot = true;
  int rc = do_upload(h, n_windows, frames, M, N, options, observations);
  h->one_shot = false;
  if (rc != MVO_HIP_OK) return rc;
  rc = do_solve(h);
  if (rc != MVO_HIP_OK) return rc;
  return do_download(h, frames, summaries);
}

int mvo_hip_refine(mvo_hip_handle h, mvo_hip_frame* frames, uint32_t M, uint32_t N,
                   const mvo_hip_options* options, const mvo_hip_observations* observations,
                   mvo_hip_summary* summary) {
  return mvo_hip_refine_batch(h, 1, frames, M, N, options, observations, summary);
}

// ---- pinned staging for the caller's gather loop (IR.cpp:116-165 writes here instead of into a
// std::list of std::vectors): the rays cross PCIe straight from where the gather put them -- no pageable
// copy by the runtime (0.195 of 0.29 ms at config 2), no second copy by the library.
int mvo_hip_staging(mvo_hip_handle h, uint32_t n_pairs, uint64_t capacity_blocks, uint64_t** pair_offsets, double** m0,
                    double** m1, uint64_t* capacity_out) {
  if (!h || !pair_offsets || !m0 || !m1) return MVO_HIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  if (h->copy_pending) {  // an upload may still be reading the block
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->copy_pending = false;
  }
  auto grow = [&](void** p, size_t old_bytes, size_t new_bytes) -> hipError_t {
    void* q = nullptr;
    hipError_t e = hipHostMalloc(&q, new_bytes, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    if (*p) {
      std::memcpy(q, *p, old_bytes);  // what the gather loop wrote so far is kept: it may ask again mid-way
      (void)hipHostFree(*p);
    }
    *p = q;
    return hipSuccess;
  };
  if (n_pairs > h->st_pairs || !h->h_st_offs) {
    const uint32_t want = n_pairs + n_pairs / 4 + 16;
    HIP_TRY(grow(reinterpret_cast<void**>(&h->h_st_offs), sizeof(uint64_t) * (h->h_st_offs ? h->st_pairs + 1 : 0),
                 sizeof(uint64_t) * ((size_t)want + 1)));
    h->st_pairs = want;
  }
  if (capacity_blocks > h->st_blocks || !h->h_st_m0) {
    const uint64_t want = capacity_blocks + capacity_blocks / 4 + 1024;
    const size_t old_bytes = sizeof(double) * 3 * (h->h_st_m0 ? h->st_blocks : 0);
    HIP_TRY(grow(reinterpret_cast<void**>(&h->h_st_m0), old_bytes, sizeof(double) * 3 * want));
    HIP_TRY(grow(reinterpret_cast<void**>(&h->h_st_m1), old_bytes, sizeof(double) * 3 * want));
    h->st_blocks = want;
  }
  *pair_offsets = h->h_st_offs;
  *m0 = h->h_st_m0;
  *m1 = h->h_st_m1;
  if (capacity_out) *capacity_out = h->st_blocks;
  return MVO_HIP_OK;
}

int mvo_hip_refine_staged(mvo_hip_handle h, mvo_hip_frame* frames, uint32_t M, uint32_t N, const mvo_hip_options* options,
                          mvo_hip_summary* summary) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (!h->h_st_offs || !h->h_st_m0) return MVO_HIP_ERR_STATE;  // mvo_hip_staging first
  if (M == 0 || N <= M) return MVO_HIP_ERR_INVALID_ARGUMENT;
  const uint32_t P = num_pairs(M, N);
  if (P > h->st_pairs || h->h_st_offs[P] > h->st_blocks) return MVO_HIP_ERR_INVALID_ARGUMENT;
  mvo_hip_observations obs;
  obs.n_pairs = P;
  obs.reserved = 0;
  obs.pair_offsets = h->h_st_offs;
  obs.m0 = h->h_st_m0;
  obs.m1 = h->h_st_m1;
  h->one_shot = true;
  int rc = do_upload(h, 1, frames, M, N, options, &obs, false, false, true);
  h->one_shot = false;
  if (rc != MVO_HIP_OK) return rc;
  rc = do_solve(h);
  if (rc != MVO_HIP_OK) return rc;
  rc = do_download(h, frames, summary);
  // the upload may point into the staging block, which the caller's next gather overwrites: it is spent
  h->uploaded = false;
  return rc;
}

int mvo_hip_eval(mvo_hip_handle h, const mvo_hip_frame* frames, uint32_t M, uint32_t N,
                 const mvo_hip_options* options, const mvo_hip_observations* observations, double* residuals,
                 double* jacobian, double* cost) {
  return mvo_hip_eval_at(h, frames, M, N, options, observations, nullptr, residuals, jacobian, cost);
}

int mvo_hip_eval_at(mvo_hip_handle h, const mvo_hip_frame* frames, uint32_t M, uint32_t N,
                    const mvo_hip_options* options, const mvo_hip_observations* observations, const double* point,
                    double* residuals, double* jacobian, double* cost) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  int rc = do_upload(h, 1, frames, M, N, options, observations);
  if (rc != MVO_HIP_OK) return rc;
  // explicit point (vec_k, t_k): replaces the uploaded initial point; queued behind the upload's copy
  if (point) HIP_TRY(hipMemcpyAsync(h->t.initx, point, sizeof(double) * 4 * M, hipMemcpyHostToDevice, h->stream));
  const uint64_t n = h->n_obs;
  const bool want_res = residuals || cost;
  if (want_res) HIP_TRY(h->d_res.reserve(sizeof(double) * (n + 1)));
  if (jacobian) HIP_TRY(h->d_jac.reserve(sizeof(double) * (n + 1) * 3 * M));
  const uint32_t blocks = (h->n_tiles + FOLD_WAVES - 1) / FOLD_WAVES;
  if (blocks) {
    hipLaunchKernelGGL(k_eval, dim3(blocks), dim3(FOLD_THREADS), 0, h->stream, make_params(h),
                       h->t.tiles, h->n_tiles, h->t.wins, h->t.pairs,
                       h->t.frames, h->t.initx, h->t.m0,
                       h->t.m1, want_res ? h->d_res.as<double>() : (double*)nullptr,
                       jacobian ? h->d_jac.as<double>() : (double*)nullptr);
    HIP_TRY(hipGetLastError());
  }
  std::vector<double> tmp;
  double* rdst = residuals;
  if (!rdst && cost) {
    tmp.resize(n);
    rdst = tmp.data();
  }
  if (rdst && n) HIP_TRY(hipMemcpyAsync(rdst, h->d_res.p, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  if (jacobian && n)
    HIP_TRY(hipMemcpyAsync(jacobian, h->d_jac.p, sizeof(double) * n * 3 * M, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (cost) {
    double c = 0.0;
    for (uint64_t i = 0; i < n; ++i) c += 0.5 * rdst[i] * rdst[i];
    *cost = c;
  }
  return MVO_HIP_OK;
}

int mvo_hip_debug_set_point(mvo_hip_handle h, uint32_t window, const double* point) {
  if (!h || !point) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (!h->uploaded) return MVO_HIP_ERR_STATE;
  if (window >= h->W) return MVO_HIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemcpyAsync(h->t.initx + (size_t)window * MVO_MAXM * 4, point, sizeof(double) * 4 * h->M,
                         hipMemcpyHostToDevice, h->stream));
  std::memcpy(&h->v_initx[(size_t)window * MVO_MAXM * 4], point, sizeof(double) * 4 * h->M);
  std::vector<double> u0;
  if (h->pass_mode) {  // the fused fold evaluates at the starting point: its unit baselines follow
    u0.resize(4 * (size_t)h->P);
    host_pair_u0(h, window, u0.data());
    HIP_TRY(hipMemcpyAsync(h->t.pair_u0 + 4 * (size_t)window * h->P, u0.data(), sizeof(double) * u0.size(),
                           hipMemcpyHostToDevice, h->stream));
  }
  HIP_TRY(hipStreamSynchronize(h->stream));  // the caller's array may change once this call returns
  return MVO_HIP_OK;
}

int mvo_hip_pair_moments(mvo_hip_handle h, double* moments) {
  if (!h || !moments) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (!h->uploaded) return MVO_HIP_ERR_STATE;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(launch_fold(h, false));
  std::vector<double> part((size_t)h->n_slots * 8 + 8);
  if (h->n_slots)
    HIP_TRY(hipMemcpyAsync(part.data(), h->d_partial.p, sizeof(double) * 8 * h->n_slots, hipMemcpyDeviceToHost,
                           h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (size_t p = 0; p < h->h_pairs.size(); ++p) {
    double S[6] = {0, 0, 0, 0, 0, 0};
    for (uint32_t sl = h->h_pairs[p].slot_begin; sl < h->h_pairs[p].slot_end; ++sl)
      for (int v = 0; v < 6; ++v) S[v] += part[(size_t)sl * 8 + v];
    std::memcpy(moments + 6 * p, S, sizeof S);
  }
  return MVO_HIP_OK;
}

int mvo_hip_debug_pair_sums(mvo_hip_handle h, double* sums) {
  if (!h || !sums) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (!h->uploaded || !h->d_pairsum.p) return MVO_HIP_ERR_STATE;
  HIP_TRY(hipSetDevice(h->device));
  HIP_TRY(hipMemcpyAsync(sums, h->d_pairsum.p, sizeof(double) * 8 * (size_t)h->W * h->P, hipMemcpyDeviceToHost,
                         h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return MVO_HIP_OK;
}

int mvo_hip_partial_begin(mvo_hip_handle h) { return h ? partial_begin(h) : MVO_HIP_ERR_INVALID_ARGUMENT; }

int mvo_hip_partial_pass(mvo_hip_handle h) { return h ? partial_pass(h) : MVO_HIP_ERR_INVALID_ARGUMENT; }

int mvo_hip_partial_advance(mvo_hip_handle h, int32_t* done) {
  return h ? partial_advance(h, done) : MVO_HIP_ERR_INVALID_ARGUMENT;
}

int mvo_hip_partial_sums(mvo_hip_handle h, void* device_buffer, uint64_t n_doubles, int direction) {
  if (!h || !device_buffer) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (!h->uploaded || !partial_buffer(h)) return MVO_HIP_ERR_STATE;
  if (n_doubles != partial_count(h)) return MVO_HIP_ERR_INVALID_ARGUMENT;
  HIP_TRY(hipSetDevice(h->device));
  if (direction == 0)
    HIP_TRY(hipMemcpyAsync(device_buffer, partial_buffer(h), sizeof(double) * n_doubles, hipMemcpyDeviceToDevice, h->stream));
  else
    HIP_TRY(hipMemcpyAsync(partial_buffer(h), device_buffer, sizeof(double) * n_doubles, hipMemcpyDeviceToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return MVO_HIP_OK;
}

uint64_t mvo_hip_partial_sums_count(mvo_hip_handle h) { return h && h->uploaded ? (uint64_t)partial_count(h) : 0; }

int mvo_hip_comm_unique_id(void* id_out) {
  if (!id_out) return MVO_HIP_ERR_INVALID_ARGUMENT;
  const MvoRccl& rc = mvo_rccl();
  if (!rc.ok) return MVO_HIP_ERR_UNSUPPORTED;  // librccl.so.1 not found
  static_assert(sizeof(ncclUniqueId) == MVO_HIP_COMM_ID_BYTES, "unique id size");
  ncclUniqueId id;
  if (rc.GetUniqueId(&id) != ncclSuccess) return MVO_HIP_ERR_DEVICE;
  std::memcpy(id_out, &id, sizeof id);
  return MVO_HIP_OK;
}

int mvo_hip_comm_init(mvo_hip_handle h, const void* unique_id, int rank, int world) {
  if (!h || !unique_id || world < 1 || rank < 0 || rank >= world) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (h->comm) return MVO_HIP_ERR_STATE;  // one communicator per handle: destroy it first
  const MvoRccl& rc = mvo_rccl();
  if (!rc.ok) return MVO_HIP_ERR_UNSUPPORTED;
  HIP_TRY(hipSetDevice(h->device));
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof id);
  const ncclResult_t r = rc.CommInitRank(&h->comm, world, id, rank);
  if (r != ncclSuccess) {
    h->last_rccl_error = r;
    h->comm = nullptr;
    return MVO_HIP_ERR_DEVICE;
  }
  h->comm_rank = rank;
  h->comm_world = world;
  return MVO_HIP_OK;
}

int mvo_hip_comm_destroy(mvo_hip_handle h) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (!h->comm) return MVO_HIP_OK;
  if (h->comm_broken) return mvo_hip_comm_abort(h);  // a collective failed: draining the stream could wait for ever
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  const ncclResult_t r = mvo_rccl().CommDestroy(h->comm);
  h->comm = nullptr;
  h->comm_world = 1;
  h->comm_rank = 0;
  return r == ncclSuccess ? MVO_HIP_OK : MVO_HIP_ERR_DEVICE;
}

int mvo_hip_comm_abort(mvo_hip_handle h) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (!h->comm) return MVO_HIP_OK;
  // no stream synchronisation: this is the way out for a rank whose peers have failed, i.e. whose stream
  // holds a collective that will never complete (ncclCommAbort ends it)
  const MvoRccl& rc = mvo_rccl();
  const ncclResult_t r = rc.CommAbort ? rc.CommAbort(h->comm) : rc.CommDestroy(h->comm);
  h->comm = nullptr;
  h->comm_world = 1;
  h->comm_rank = 0;
  h->comm_broken = false;
  h->solved = false;       // whatever the aborted solve left behind is not a result
  h->done_ready = false;
  return r == ncclSuccess ? MVO_HIP_OK : MVO_HIP_ERR_DEVICE;
}

int mvo_hip_solve_partitioned(mvo_hip_handle h) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  return do_solve_partitioned(h);
}

int mvo_hip_set_profiling(mvo_hip_handle h, int enabled) {
  if (!h) return MVO_HIP_ERR_INVALID_ARGUMENT;
  h->profiling = enabled != 0;
  return MVO_HIP_OK;
}

int mvo_hip_set_trace(mvo_hip_handle h, int32_t capacity) {
  if (!h || capacity < 0) return MVO_HIP_ERR_INVALID_ARGUMENT;
  h->trace_cap = capacity;
  return MVO_HIP_OK;
}

int mvo_hip_get_trace(mvo_hip_handle h, mvo_hip_iteration* rows, int32_t capacity, int32_t* n_rows) {
  if (!h || !n_rows) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (!h->uploaded || !h->solved) return MVO_HIP_ERR_STATE;
  const int32_t produced = h->h_results[0].iterations + 1;
  *n_rows = produced;
  int32_t n = produced < h->trace_cap ? produced : h->trace_cap;
  if (n > capacity) n = capacity;
  if (rows && n > 0) {
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpy(rows, h->d_trace.p, sizeof(mvo_hip_iteration) * (size_t)n, hipMemcpyDeviceToHost));
  }
  return MVO_HIP_OK;
}

int mvo_hip_get_timing(mvo_hip_handle h, mvo_hip_timing* out) {
  if (!h || !out) return MVO_HIP_ERR_INVALID_ARGUMENT;
  *out = h->timing;
  return MVO_HIP_OK;
}

double mvo_hip_reverse_scale(double length, double highest_length, double lowest_length) {
  return host_reverse_scale(length, highest_length, lowest_length);
}

double mvo_hip_scale(double t, double highest_length, double lowest_length) {
  return host_scale(t, highest_length, lowest_length);
}

uint32_t mvo_hip_num_pairs(uint32_t M, uint32_t N) {
  if (M == 0 || N <= M) return 0;
  return num_pairs(M, N);
}

const char* mvo_hip_status_string(int status) {
  switch (status) {
    case MVO_HIP_OK: return "ok";
    case MVO_HIP_ERR_INVALID_ARGUMENT: return "invalid argument";
    case MVO_HIP_ERR_UNSUPPORTED: return "unsupported problem shape (numberToRefine > 5, too many frames or observations)";
    case MVO_HIP_ERR_OUT_OF_MEMORY: return "out of device memory";
    case MVO_HIP_ERR_DEVICE: return "HIP runtime error";
    case MVO_HIP_ERR_NO_DEVICE: return "no HIP device (libmvo_hip has no CPU fallback)";
    case MVO_HIP_ERR_STATE: return "no problem uploaded";
    default: return "unknown status";
  }
}

#ifdef MVO_LM_STAMPS
// diagnostic build: phase cycle counters of k_lm (window 0); [0] = iterations counted
int mvo_hip_debug_stamps(unsigned long long* out8, int reset) {
  if (reset) {
    unsigned long long z[32] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(mvo_lm_stamps), z, sizeof z) == hipSuccess ? 0 : 4;
  }
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(mvo_lm_stamps), 128) == hipSuccess ? 0 : 4;
}
// slots 16..23: the persistent robust solver's phases (MVO_RSTAMP, mvo_lm.h)
int mvo_hip_debug_stamps_persistent(unsigned long long* out8) {
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(mvo_lm_stamps), 64, 128) == hipSuccess ? 0 : 4;
}
// slots 24..31: k_lm4's phases (workgroup 0), mvo_lm4.h
int mvo_hip_debug_stamps_lm4(unsigned long long* out8) {
  return hipMemcpyFromSymbol(out8, HIP_SYMBOL(mvo_lm_stamps), 64, 192) == hipSuccess ? 0 : 4;
}
#endif

const char* mvo_hip_version(void) { return "libmvo_hip 0.1 (gfx950)"; }

long mvo_hip_debug_check_guards(void) {
  if (!alloc_poison()) return -1;
  if (hipDeviceSynchronize() != hipSuccess) return -2;
  AllocRegistry& r = alloc_registry();
  std::lock_guard<std::mutex> lock(r.mu);
  long bad = r.violations;
  for (const auto& kv : r.live)
    if (!guard_intact(kv.first, kv.second)) ++bad;
  return bad;
}

int mvo_hip_debug_poison_lds(int device_ordinal) {
  hipDeviceProp_t prop;
  if (hipSetDevice(device_ordinal) != hipSuccess || hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess)
    return MVO_HIP_ERR_DEVICE;
  const int lds_bytes = 64 * 1024;  // two such workgroups share a CU's 160 KB: several rounds cover all of it
  hipLaunchKernelGGL(k_poison_lds, dim3((unsigned)prop.multiProcessorCount * 16u), dim3(256), lds_bytes, 0,
                     lds_bytes / 8);
  if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) return MVO_HIP_ERR_DEVICE;
  return MVO_HIP_OK;
}

}  // extern "C"

// =========================================================================================
// mvo_hip_refine_noalloc: the one entry point that allocates nothing (include/mvo_hip.h).
// Same kernels as mvo_hip_refine's single-launch path (k_lm with the window's own observation
// passes); everything those kernels keep beyond the caller's inputs and outputs -- window and pair
// tables, the initial point, per-pair sums, LM state, result record, status / done words -- lives
// in ONE file-scope device array.  Host code never passes the array's name to a kernel: its
// address on a device is resolved once per device with hipGetSymbolAddress and kept.
// =========================================================================================
namespace {
constexpr uint32_t kNoallocMaxPairs = MVO_MAXM * (MVO_HIP_MAX_FRAMES - 1) - (MVO_MAXM * (MVO_MAXM - 1)) / 2;
constexpr size_t na_align(size_t x) { return (x + 255) & ~(size_t)255; }
constexpr size_t kNaWin = 0;                                                    // WinDev
constexpr size_t kNaCtl = na_align(kNaWin + sizeof(WinDev));                    // int32: status, done, bad
constexpr size_t kNaResult = na_align(kNaCtl + 4 * 64);                         // LmResult
constexpr size_t kNaState = na_align(kNaResult + sizeof(LmResult));             // LmState
constexpr size_t kNaInitX = na_align(kNaState + sizeof(LmState));               // MVO_MAXM x (vec, t)
constexpr size_t kNaPairs = na_align(kNaInitX + sizeof(double) * 4 * MVO_MAXM); // PairDev[P]
constexpr size_t kNaSums = na_align(kNaPairs + sizeof(PairDev) * kNoallocMaxPairs);  // 8 doubles per pair
constexpr size_t kNaEnd = na_align(kNaSums + sizeof(double) * 8 * kNoallocMaxPairs);
static_assert(kNaEnd <= MVO_HIP_NOALLOC_SCRATCH_BYTES, "scratch layout exceeds the declared size");
constexpr int kNaMaxDevices = 64;
}  // namespace

static __device__ __align__(256) unsigned char scratch[MVO_HIP_NOALLOC_SCRATCH_BYTES];

namespace {
struct NoallocDevice {
  unsigned char* ptr = nullptr;  // this device's copy of `scratch`
};
NoallocDevice g_noalloc[kNaMaxDevices];
std::mutex g_noalloc_mu;

// once per device, with that device current
hipError_t noalloc_setup(int device, NoallocDevice** out) {
  if (device < 0 || device >= kNaMaxDevices) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lock(g_noalloc_mu);
  NoallocDevice& d = g_noalloc[device];
  if (!d.ptr) {
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    d.ptr = static_cast<unsigned char*>(p);
  }
  *out = &d;
  return hipSuccess;
}

}  // namespace

hipError_t mvo_noalloc_resolve(int device) {
  NoallocDevice* d = nullptr;
  return noalloc_setup(device, &d);
}

namespace {
// IR.cpp:348-368 on the device (one clamp step instead of the recursion: the clamped value lies
// strictly inside the interval unless the interval itself is degenerate)
__device__ double d_reverse_scale(double length, double max_len, double min_len, int& clamped) {
  const double eps = DBL_EPSILON;
  clamped = 0;
  if (length <= min_len) {
    length = min_len + (length * eps) + eps;
    clamped = 1;
  } else if (length >= max_len) {
    length = max_len - (length * eps + eps);
    clamped = 2;
  }
  return -1.0 * log((max_len - length) / (length - min_len));
}

// tables of the one window, built where they are used: what do_upload builds on the host
__global__ __launch_bounds__(256) void k_noalloc_prep(unsigned char* sc, const double* __restrict__ frames,
                                                      const uint64_t* __restrict__ offs, uint32_t M, uint32_t N,
                                                      uint32_t P, uint64_t n_blocks, double max_len, double min_len) {
  __shared__ int bad_s, clamp_s;
  const int tid = threadIdx.x;
  if (tid == 0) bad_s = clamp_s = 0;
  __syncthreads();
  int bad = 0;
  for (uint32_t p = tid; p < P; p += 256)
    if (offs[p + 1] < offs[p] || offs[p + 1] > n_blocks) bad = 1;
  if (tid == 0 && (offs[0] != 0 || offs[P] != n_blocks)) bad = 1;
  if (bad) bad_s = 1;
  __syncthreads();
  const bool ok = bad_s == 0;
  PairDev* pairs = reinterpret_cast<PairDev*>(sc + kNaPairs);
  for (uint32_t p = tid; p < P; p += 256) {
    int a, b;
    pair_ab((int)N, (int)M, (int)p, a, b);
    PairDev pd;
    pd.slot_begin = p;
    pd.slot_end = p + 1;
    pd.a = (uint32_t)a;
    pd.b = (uint32_t)b;
    pd.src_start = ok ? (uint32_t)offs[p] : 0u;
    pd.count = ok ? (uint32_t)(offs[p + 1] - offs[p]) : 0u;  // inconsistent offsets: nothing is read
    pairs[p] = pd;
  }
  if (tid < (int)M) {
    double* x = reinterpret_cast<double*>(sc + kNaInitX) + 4 * tid;
    const double* f = frames + MVO_FRAME_DOUBLES * tid;
    x[0] = f[9];
    x[1] = f[10];
    x[2] = f[11];
    int clamped;
    x[3] = d_reverse_scale(f[12], max_len, min_len, clamped);  // IR.cpp:102
    if (clamped) atomicOr(&clamp_s, 1 << (2 * tid + (clamped - 1)));
    // the frames are device memory: what mvo_hip_refine's validate() refuses on the host (an infinite or
    // NaN length, on which the reference's clamp recursion never ends) is reported as BAD_INPUT here
    if (!isfinite(f[12]) || !isfinite(x[3])) atomicOr(&bad_s, 1);
  }
  __syncthreads();
  if (tid == 0) {
    WinDev w{};
    w.frame_off = 0;
    w.pair_off = 0;
    w.N = N;
    w.M = M;
    w.P = P;
    *reinterpret_cast<WinDev*>(sc + kNaWin) = w;
    int32_t* ctl = reinterpret_cast<int32_t*>(sc + kNaCtl);
    ctl[0] = 0;         // window status
    ctl[16] = 0;        // done counter
    ctl[32] = bad_s;    // inconsistent pair_offsets
    ctl[48] = clamp_s;  // which initial lengths reverseScale clamped
  }
}

// IR.cpp:240-246 on the device: refined vec / scale into the caller's frames, summary out
__global__ void k_noalloc_finish(const unsigned char* sc, double* __restrict__ frames, uint32_t M,
                                 mvo_hip_summary* __restrict__ summary) {
  const LmResult* r = reinterpret_cast<const LmResult*>(sc + kNaResult);
  const int32_t* ctl = reinterpret_cast<const int32_t*>(sc + kNaCtl);
  const bool bad = ctl[32] != 0 || ctl[0] == 0;  // bad offsets, or the window never terminated
  const int k = threadIdx.x;
  if (!bad && k < (int)M) {
    double* f = frames + MVO_FRAME_DOUBLES * k;
    f[9] = r->vec[k][0];
    f[10] = r->vec[k][1];
    f[11] = r->vec[k][2];
    f[12] = r->scale[k];
  }
  if (k == 0 && summary) {
    mvo_hip_summary s;
    s.termination = bad ? MVO_HIP_FAILURE : r->termination;
    s.reason = bad ? MVO_HIP_REASON_BAD_INPUT : r->reason;
    s.iterations = bad ? 0 : r->iterations;
    s.num_successful_steps = bad ? 0 : r->nsucc;
    s.num_unsuccessful_steps = bad ? 0 : r->nunsucc;
    s.clamped_scales = ctl[48];
    s.initial_cost = bad ? 0.0 : r->initial_cost;
    s.final_cost = bad ? 0.0 : r->final_cost;
    *summary = s;
  }
}
}  // namespace

int mvo_hip_refine_noalloc(int device_ordinal, void* stream, mvo_hip_frame* frames_device, uint32_t M, uint32_t N,
                           const mvo_hip_options* options, const uint64_t* pair_offsets_device,
                           const double* m0_device, const double* m1_device, uint64_t n_blocks,
                           mvo_hip_summary* summary_device) {
  if (!frames_device || !options || !pair_offsets_device) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (M == 0 || N <= M) return MVO_HIP_ERR_INVALID_ARGUMENT;   // IR.cpp:29-30
  if (M > MVO_HIP_MAX_REFINE) return MVO_HIP_ERR_UNSUPPORTED;  // IR.cpp:450
  if (N > MVO_HIP_MAX_FRAMES) return MVO_HIP_ERR_UNSUPPORTED;  // beyond the scratch array's pair tables
  if (n_blocks > MVO_HIP_NOALLOC_MAX_BLOCKS) return MVO_HIP_ERR_UNSUPPORTED;
  if (n_blocks > 0 && (!m0_device || !m1_device)) return MVO_HIP_ERR_INVALID_ARGUMENT;
  if (options->max_num_iterations < 0 || !(options->highest_length > options->lowest_length) ||
      !std::isfinite(options->highest_length) || !std::isfinite(options->lowest_length))
    return MVO_HIP_ERR_INVALID_ARGUMENT;
  {
    // as validate() for mvo_hip_refine: reverseScale's clamp (IR.cpp:350-363) must land strictly inside
    // (lowest, highest) -- the reference would recurse forever otherwise
    const double eps = std::numeric_limits<double>::epsilon();
    const double lo = options->lowest_length + (options->lowest_length * eps) + eps;
    const double hi = options->highest_length - (options->highest_length * eps + eps);
    if (!(lo > options->lowest_length && lo < options->highest_length && hi < options->highest_length &&
          hi > options->lowest_length))
      return MVO_HIP_ERR_INVALID_ARGUMENT;
  }
  if (hipSetDevice(device_ordinal) != hipSuccess) return MVO_HIP_ERR_DEVICE;
  NoallocDevice* dev = nullptr;
  if (noalloc_setup(device_ordinal, &dev) != hipSuccess) return MVO_HIP_ERR_DEVICE;
  unsigned char* sc = dev->ptr;  // never the array's name: that is a host-side stand-in
  hipStream_t s = static_cast<hipStream_t>(stream);
  const uint32_t P = num_pairs(M, N);

  SolveParams prm{};
  prm.ftol = options->function_tolerance;
  prm.gtol = options->gradient_tolerance;
  prm.ptol = options->parameter_tolerance;
  prm.min_len = options->lowest_length;
  prm.max_len = options->highest_length;
  prm.max_it = options->max_num_iterations;
  prm.use_loss = options->use_loss_function ? 1 : 0;
  prm.fix_length = options->fix_length ? 1 : 0;
  prm.pass_mode = (options->use_loss_function || (options->assembly_mode & MVO_HIP_ASSEMBLY_PASSES)) ? 1 : 0;
  prm.trace_cap = 0;
  prm.inline_eval = 1;  // the window's workgroup runs its own observation passes: one launch
  prm.fresh = 1;        // k_lm starts from a zero state of its own: nothing of the last call is read
  prm.win_base = 0;
  prm.M = (int32_t)M;
  prm.N = (int32_t)N;
  prm.P = (int32_t)P;

  const double* fr = reinterpret_cast<const double*>(frames_device);
  hipLaunchKernelGGL(k_noalloc_prep, dim3(1), dim3(256), 0, s, sc, fr, pair_offsets_device, M, N, P, n_blocks,
                     prm.max_len, prm.min_len);
  if (hipGetLastError() != hipSuccess) return MVO_HIP_ERR_DEVICE;

  const bool small = P <= 64;
  const WinDev* wins = reinterpret_cast<const WinDev*>(sc + kNaWin);
  const PairDev* pairs = reinterpret_cast<const PairDev*>(sc + kNaPairs);
  double* sums = reinterpret_cast<double*>(sc + kNaSums);
  int32_t* ctl = reinterpret_cast<int32_t*>(sc + kNaCtl);
  auto launch = [&](auto kernel, int nt) {
    hipLaunchKernelGGL(kernel, dim3(1), dim3(nt), 0, s, prm, 1, wins, fr,
                       reinterpret_cast<const double*>(sc + kNaInitX), pairs, (const double*)sums, m0_device,
                       m1_device, sums, reinterpret_cast<LmState*>(sc + kNaState),
                       reinterpret_cast<LmResult*>(sc + kNaResult), ctl, (TraceRec*)nullptr,
                       ctl + 16, (const double*)nullptr, (const double*)nullptr, (double*)nullptr, RobustArgs{});
  };
  if (small) launch(lm_kernel<64>(lm_spec(prm, false)), 64);
  else launch(lm_kernel<MVO_LM_NT_LARGE>(lm_spec(prm, false)), MVO_LM_NT_LARGE);
  if (hipGetLastError() != hipSuccess) return MVO_HIP_ERR_DEVICE;

  hipLaunchKernelGGL(k_noalloc_finish, dim3(1), dim3(64), 0, s, (const unsigned char*)sc,
                     reinterpret_cast<double*>(frames_device), M, summary_device);
  if (hipGetLastError() != hipSuccess) return MVO_HIP_ERR_DEVICE;
  return MVO_HIP_OK;
}
