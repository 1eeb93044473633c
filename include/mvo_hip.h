/*
 * mvo_hip.h -- C ABI of libmvo_hip, the MI355X (gfx950) implementation of the reference's
 * sliding-window refinement solver.
 *
 * Drop-in boundary: the reference has no plugin/FFI layer; its operator API for this path is
 * the C++ member
 *
 *   void IterativeRefinement::refine(std::vector<RefinementFrame>&, unsigned numberToRefine,
 *        unsigned numberToNote, int maxNumthreads, int maxNumIterations,
 *        double functionTolerance, double gradientTolerance, double parameterTolerance,
 *        bool useLossFunction, double lowestLength, double highestLength, const Frame&)
 *        -- mvo/src/algorithms/IterativeRefinement.hpp:181-192, called from
 *           mvo/src/pipeline/Refiner.cpp:85-92.
 *
 * Everything below is what a binding for that call needs: plain pointers and sizes, no C++
 * types, no exceptions, no logging.  The header is C99 and pedantic-clean C++11 (the reference
 * compiles with -std=c++11 -Wall -Werror -pedantic -Wextra, mvo/CMakeLists.txt:4,71).
 *
 * Conventions (identical to the reference):
 *   - frames[i], i = 0 newest ... N-1 oldest (Refiner.cpp:62-68); N = numberToNote,
 *     M = numberToRefine, 1 <= M < N (IterativeRefinement.cpp:28-30), M <= 5
 *     (IterativeRefinement.cpp:384-451, default: assert(false)).
 *   - residual blocks are grouped by frame pair in the reference's loop order
 *     "for a in [0,M): for b in [a,N-1)" (IterativeRefinement.cpp:116,134); pair p owns blocks
 *     [pair_offsets[p], pair_offsets[p+1]); m0 = unit ray in frame a, m1 = unit ray in frame
 *     b+1 (IterativeRefinement.cpp:152-154).
 */
#ifndef MVO_HIP_H
#define MVO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVO_HIP_MAX_REFINE 5    /* IterativeRefinement.cpp:384-451 */
#define MVO_HIP_MAX_FRAMES 256  /* mvo/cfg/mvo.cfg:43 allows numberToNote <= 100 */

/* status codes returned by every entry point */
#define MVO_HIP_OK 0
#define MVO_HIP_ERR_INVALID_ARGUMENT 1 /* the reference's assert()s, IterativeRefinement.cpp:28-30,379 */
#define MVO_HIP_ERR_UNSUPPORTED 2      /* M > 5: the reference's "default: assert(false)" */
#define MVO_HIP_ERR_OUT_OF_MEMORY 3
#define MVO_HIP_ERR_DEVICE 4           /* a HIP runtime call failed */
#define MVO_HIP_ERR_NO_DEVICE 5        /* no gfx950 device: there is NO CPU fallback */
#define MVO_HIP_ERR_STATE 6            /* solve without a prior upload; download / get_trace without a completed solve */

/* IterativeRefinement::RefinementFrame (IterativeRefinement.hpp:171-175).
 * R: cv::Matx33d row-major (IterativeRefinement.cpp:320-324); vec: unit baseline direction
 * frame i -> i+1 in world axes; scale: baseline length. 104 bytes, no padding. */
typedef struct mvo_hip_frame {
  double R[9];
  double vec[3];
  double scale;
} mvo_hip_frame;

/* The scalar arguments of refine() (IterativeRefinement.hpp:183-191) plus fixLength, which the
 * reference reads via nowFrame.getParameters() (IterativeRefinement.cpp:105).
 * Tolerances are the already-converted 10^-x values (Refiner.cpp:72-74). */
typedef struct mvo_hip_options {
  int32_t max_num_iterations;  /* maxNumIterations   -> Solver::Options::max_num_iterations */
  int32_t num_threads;         /* maxNumthreads: accepted for signature parity, unused on GPU */
  double function_tolerance;
  double gradient_tolerance;
  double parameter_tolerance;
  int32_t use_loss_function;   /* ceres::CauchyLoss(0.5), IterativeRefinement.cpp:85-88 */
  int32_t fix_length;          /* SetParameterBlockConstant(scale), IterativeRefinement.cpp:105-107 */
  double lowest_length;        /* lowestLength  (MIN_LEN of scaleTemplated) */
  double highest_length;       /* highestLength (MAX_LEN of scaleTemplated) */
  /* library extension, 0 = reference behaviour.
   * MVO_HIP_ASSEMBLY_AUTO  : no loss -> per-pair second moments are folded once and the LM loop
   *                          never touches observations again; loss -> one pass per evaluation.
   * MVO_HIP_ASSEMBLY_PASSES: always one observation pass per evaluation (what Ceres does). */
  int32_t assembly_mode;
  int32_t reserved;
} mvo_hip_options;

#define MVO_HIP_ASSEMBLY_AUTO 0
#define MVO_HIP_ASSEMBLY_PASSES 1 /* bit 0 */
#define MVO_HIP_ASSEMBLY_TILED 2  /* bit 1: always use the chip-wide tiled observation passes; by default
                                     small or many windows are evaluated by their own LM workgroup */
/* Batches of windows with at most 64 frame pairs, no loss: one window per wavefront (k_lm<64>) or four
 * (k_lm4, shares the LM step between four windows: 1.35 x the time of one wavefront for 4 x the windows).
 * By default the library takes whichever needs fewer rounds of the chip's 1024 wavefront slots (four per
 * wavefront from 1025 windows up); these two bits force one or the other.  Same results to rounding. */
#define MVO_HIP_ASSEMBLY_ONE_PER_WAVE 4  /* bit 2 */
#define MVO_HIP_ASSEMBLY_FOUR_PER_WAVE 8 /* bit 3 */

/* What IterativeRefinement.cpp:111-183 gathers from nowFrame: replaces the
 * std::list<std::vector<Eigen::Vector3d>> "features" (IterativeRefinement.cpp:111,147-165). */
typedef struct mvo_hip_observations {
  uint32_t n_pairs;              /* must equal M(N-1) - M(M-1)/2 */
  uint32_t reserved;
  const uint64_t *pair_offsets;  /* n_pairs + 1 entries, CSR over residual blocks */
  const double *m0;              /* 3 doubles per residual block (x,y,z interleaved), unit norm */
  const double *m1;              /* likewise, ray in frame b+1 */
} mvo_hip_observations;

/* ceres::Solver::Summary subset (IterativeRefinement.cpp:231-237,248). */
#define MVO_HIP_CONVERGENCE 0
#define MVO_HIP_NO_CONVERGENCE 1
#define MVO_HIP_FAILURE 2

#define MVO_HIP_REASON_NONE 0
#define MVO_HIP_REASON_MAX_ITERATIONS 1
#define MVO_HIP_REASON_GRADIENT_TOLERANCE 2
#define MVO_HIP_REASON_PARAMETER_TOLERANCE 3
#define MVO_HIP_REASON_FUNCTION_TOLERANCE 4
#define MVO_HIP_REASON_MIN_RADIUS 5
#define MVO_HIP_REASON_INVALID_STEPS 6
#define MVO_HIP_REASON_LINEAR_SOLVER 7
#define MVO_HIP_REASON_EVALUATION_FAILED 8 /* non-finite residuals at the initial point: FAILURE before any step */
#define MVO_HIP_REASON_BAD_INPUT 9         /* mvo_hip_refine_noalloc: device-side pair_offsets inconsistent */

typedef struct mvo_hip_summary {
  int32_t termination;           /* MVO_HIP_CONVERGENCE / NO_CONVERGENCE / FAILURE */
  int32_t reason;
  int32_t iterations;            /* index of the last finalized Ceres iteration */
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  /* which initial lengths reverseScale had to clamp into (lowestLength, highestLength) before the
   * solve: bit 2k = frame k was <= lowestLength (the ROS_WARN of IterativeRefinement.cpp:357),
   * bit 2k+1 = frame k was >= highestLength (:363).  The library never logs; the adapter does. */
  int32_t clamped_scales;
  double initial_cost;
  double final_cost;
} mvo_hip_summary;

/* one row per ceres::IterationSummary; optional debugging aid */
typedef struct mvo_hip_iteration {
  double cost;
  double cost_change;
  double gradient_max_norm;
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
  int32_t step_is_valid;
  int32_t step_is_successful;
} mvo_hip_iteration;

/* device-side timing of the last mvo_hip_solve(), measured with HIP events on the library's
 * own stream (the reference's only instrumentation is DUR_REF, Refiner.cpp:40,118) */
typedef struct mvo_hip_timing {
  double upload_seconds;          /* host wall time of the last upload (tables + H2D) */
  double solve_seconds;           /* host wall time of the last solve, launch -> results on host */
  double device_seconds;          /* first kernel start -> last kernel end */
  double assembly_seconds;        /* sum over launches of the observation-pass kernels */
  int32_t assembly_launches;
  int32_t lm_launches;
  uint64_t residual_blocks;       /* per launch of the assembly kernel */
  uint64_t assembly_bytes;        /* algorithmic bytes per launch of the dominant assembly kernel */
  double assembly_first_seconds;  /* the fold launch alone (reads m0,m1; writes c in pass mode) */
} mvo_hip_timing;

typedef struct mvo_hip_context *mvo_hip_handle;

/* Owns one HIP stream, the device buffers and pinned staging memory; not re-entrant per handle
 * (the reference calls refine() from one thread only, mvo/src/mvo.cpp:37).  The capacities are
 * hints: buffers grow on demand.  Fails with MVO_HIP_ERR_NO_DEVICE when no GPU is present. */
int mvo_hip_create(int device_ordinal, uint32_t max_windows, uint32_t max_frames,
                   uint64_t max_observations, mvo_hip_handle *out);
void mvo_hip_destroy(mvo_hip_handle h);

/* Replaces IterativeRefinement::refine (IterativeRefinement.cpp:15-250) from the solver set-up
 * (:63-109) to the write-back (:240-246): frames[0..M) receive the refined vec and scale.  After
 * FAILURE (summary.termination == MVO_HIP_FAILURE) Ceres leaves the caller's parameter arrays at
 * their INITIAL values (the summary is not usable) and the reference writes those back
 * (:235-246): frames[0..M) then receive vec_0 and s(reverseScale(scale_0)), whatever progress the
 * solve made before it failed.  summary may be NULL. */
int mvo_hip_refine(mvo_hip_handle h, mvo_hip_frame *frames, uint32_t M, uint32_t N,
                   const mvo_hip_options *options, const mvo_hip_observations *observations,
                   mvo_hip_summary *summary);

/* The same call with the gathering loop (IterativeRefinement.cpp:116-165) writing straight into PINNED
 * staging memory of the handle instead of into the reference's std::list<std::vector<Vector3d>> "features"
 * (IterativeRefinement.cpp:111,147-165): the rays then cross PCIe from where the gather put them -- no
 * pageable copy by the HIP runtime (0.195 of 0.29 ms at BASELINE configs[1]), no second copy by the library.
 * mvo_hip_staging returns room for n_pairs + 1 offsets and capacity_blocks unit rays per side (3 doubles per
 * block); the block only grows and keeps what it held when it does, so a gather loop that does not know its
 * block count up front may ask again mid-way with a larger capacity (*capacity_out: blocks that fit now).
 * The pointers stay valid until the next mvo_hip_staging call that grows the block, or mvo_hip_destroy.
 * mvo_hip_refine_staged is mvo_hip_refine on what the staging block holds: pair_offsets[0 .. P] as written
 * by the caller (P = M(N-1) - M(M-1)/2), rays for blocks [0, pair_offsets[P]).  Not re-entrant per handle. */
int mvo_hip_staging(mvo_hip_handle h, uint32_t n_pairs, uint64_t capacity_blocks, uint64_t **pair_offsets,
                    double **m0, double **m1, uint64_t *capacity_out);
int mvo_hip_refine_staged(mvo_hip_handle h, mvo_hip_frame *frames, uint32_t M, uint32_t N,
                          const mvo_hip_options *options, mvo_hip_summary *summary);

/* n_windows independent refine() calls of identical (M, N) in one launch sequence
 * (multi-sequence replay; no reference counterpart beyond calling refine() in a loop).
 * frames: n_windows * N records, window-major.  observations / summaries: one per window. */
int mvo_hip_refine_batch(mvo_hip_handle h, uint32_t n_windows, mvo_hip_frame *frames, uint32_t M,
                         uint32_t N, const mvo_hip_options *options,
                         const mvo_hip_observations *observations, mvo_hip_summary *summaries);

/* Split-phase form of the same call, so that callers (and bench.py) can keep the observations
 * resident in HBM: upload = IterativeRefinement.cpp:36-43,111-183 (packing + gathering),
 * solve = :63-109,231-233, download = :240-246.  solve() may be repeated; every solve restarts
 * from the uploaded initial values. */
int mvo_hip_upload(mvo_hip_handle h, uint32_t n_windows, const mvo_hip_frame *frames, uint32_t M,
                   uint32_t N, const mvo_hip_options *options,
                   const mvo_hip_observations *observations);
int mvo_hip_solve(mvo_hip_handle h);
int mvo_hip_download(mvo_hip_handle h, mvo_hip_frame *frames, mvo_hip_summary *summaries);

/* Observation-partitioned solve of ONE large window over several GPUs (no reference counterpart:
 * the reference is single-process).  Every rank uploads the same frames and
 * options and ITS share of each pair's residual blocks.  What the ranks exchange is additive over
 * ranks; the caller all-reduces it (RCCL/torch.distributed) between the calls below, after which
 * every rank takes the identical LM step.  Each call drains the library's stream before returning.
 *   no loss : begin -> [all-reduce] -> advance (whole LM loop in one launch, done = 1).
 *             Exchanged: the per-pair sums, 8 doubles per pair (S xx,xy,xz,yy,yz,zz, sum r^2, unused),
 *             ONCE per solve -- S_p does not depend on the evaluation point.
 *   loss    : begin -> { [all-reduce] -> advance -> done ? stop : pass } ...
 *             begin folds the observations and is the evaluation pass at the starting point; `pass` is
 *             the pass at the candidate `advance` published.  Exchanged: the CONTRACTED form, the class
 *             totals of the pairs' projected moments P S P, P S u, 1/2 sum rho (MVO_HIP_CLASS_DOUBLES =
 *             352 doubles per window and evaluation, linear in every rank's share), so that no rank
 *             repeats the per-pair stage of the step.
 * mvo_hip_partial_sums_count returns the number of doubles exchanged for the current upload;
 * partial_sums copies them to (direction 0) or from (direction 1) a caller-owned DEVICE buffer. */
#define MVO_HIP_CLASS_DOUBLES 352
int mvo_hip_partial_begin(mvo_hip_handle h);
int mvo_hip_partial_pass(mvo_hip_handle h);
int mvo_hip_partial_advance(mvo_hip_handle h, int32_t *done);
uint64_t mvo_hip_partial_sums_count(mvo_hip_handle h);
int mvo_hip_partial_sums(mvo_hip_handle h, void *device_buffer, uint64_t n_doubles, int direction);

/* The same partitioned solve with the collective INSIDE the library (north_star: "partial
 * normal-equation blocks are combined with an RCCL all-reduce over xGMI"; no reference counterpart).
 * One process per GPU: rank 0 calls mvo_hip_comm_unique_id and ships the MVO_HIP_COMM_ID_BYTES bytes to
 * the other ranks by whatever means the application has (MPI, a file, a socket); every rank then calls
 * mvo_hip_comm_init on its handle (ncclCommInitRank on the handle's device: collective, blocks until
 * all ranks have arrived).  After mvo_hip_upload of this rank's shard (same frames and options on every
 * rank), mvo_hip_solve_partitioned runs fold / pass -> this rank's sums -> ncclAllReduce(sum; 8 fp64 per
 * pair once per solve without a loss, MVO_HIP_CLASS_DOUBLES fp64 per window and evaluation with one) ->
 * LM step entirely on the handle's stream -- no host round trip per pass; results as after
 * mvo_hip_solve, identical on every rank (mvo_hip_download).  Every rank must make the same calls in
 * the same order.  Fail-stop: a rank whose HIP or RCCL call fails returns an error while the others wait
 * in their next collective -- tear the job down, as after any failed collective: mvo_hip_comm_abort
 * (ncclCommAbort, no wait for the stream; callable from another thread than the one stuck in the solve)
 * ends this rank's pending collectives and releases the communicator, and mvo_hip_comm_destroy does the
 * same by itself on a handle whose own collective call failed.  After an abort the handle holds no
 * result (mvo_hip_download: MVO_HIP_ERR_STATE) and may be given a new communicator.  librccl.so.1 is opened on the first of these calls (MVO_HIP_ERR_UNSUPPORTED when it
 * cannot be found), so single-GPU users never load it.  mvo_hip_comm_destroy (or mvo_hip_destroy)
 * releases the communicator. */
#define MVO_HIP_COMM_ID_BYTES 128
int mvo_hip_comm_unique_id(void *id_out);
int mvo_hip_comm_init(mvo_hip_handle h, const void *unique_id, int rank, int world);
int mvo_hip_comm_destroy(mvo_hip_handle h);
int mvo_hip_comm_abort(mvo_hip_handle h);
int mvo_hip_solve_partitioned(mvo_hip_handle h);

/* Debug/parity entry point: the raw residual r_i and the raw local Jacobian row of every
 * residual block at the initial point -- what ceres::Problem::Evaluate would return for the
 * problem built at IterativeRefinement.cpp:95-183 with apply_loss_function = false.
 * jacobian is row-major n_obs x 3M, columns of refined frame k are (d/da, d/db, d/dt) with
 * (a,b) the ParametrizedBaseLine chart (IterativeRefinement.cpp:334-346) and t the sigmoid
 * parameter of scaleTemplated (:298-309).  Any output pointer may be NULL. */
int mvo_hip_eval(mvo_hip_handle h, const mvo_hip_frame *frames, uint32_t M, uint32_t N,
                 const mvo_hip_options *options, const mvo_hip_observations *observations,
                 double *residuals, double *jacobian, double *cost);

/* The same at an explicit point: point[4 * k .. 4 * k + 3] = (vec_k[3], t_k) for k < M replaces the
 * initial point (vec from the frames, t = reverseScale(scale)); NULL = mvo_hip_eval.  Lets a test
 * reach the saturated branch of scaleTemplated (IterativeRefinement.cpp:301-307: exp(-t) infinite =>
 * s = lowestLength, derivative 0), which reverseScale can never produce (|t| <~ 36). */
int mvo_hip_eval_at(mvo_hip_handle h, const mvo_hip_frame *frames, uint32_t M, uint32_t N,
                    const mvo_hip_options *options, const mvo_hip_observations *observations,
                    const double *point, double *residuals, double *jacobian, double *cost);

/* Test aid: replaces the starting point of window `window` of the last upload (for the solves that
 * follow, until the next upload) by point[4 * k .. 4 * k + 3] = (vec_k[3], t_k): the only way to
 * start the LM loop on the saturated branch of scaleTemplated.  A FAILURE hands this point back. */
int mvo_hip_debug_set_point(mvo_hip_handle h, uint32_t window, const double *point);

/* Debug/parity: per-pair second moments S_p = sum_i c_i c_i^T (xx,xy,xz,yy,yz,zz) of the last
 * upload, window-major, n_windows * n_pairs * 6 doubles. */
int mvo_hip_pair_moments(mvo_hip_handle h, double *moments);

/* Debug/parity: the per-pair sums the LM loop last read (8 doubles per pair: S xx,xy,xz,yy,yz,zz,
 * sum rho or sum r^2, sum w r^2), n_windows * n_pairs * 8 doubles to host memory.  After a solve of the
 * tiled path these are what the observation pass's own tail (or, with MVO_HIP_NO_TAILS set when the handle
 * was created, a k_reduce_slots launch) added up from the workgroups' slots -- the tests compare the two
 * bit for bit.  Replaces nothing in the reference (its accumulation is Ceres' AddResidualBlock loop,
 * IterativeRefinement.cpp:369-455). */
int mvo_hip_debug_pair_sums(mvo_hip_handle h, double *sums);

/* Iteration trace of window 0 of the next solves: capacity rows are recorded (0 disables). */
int mvo_hip_set_trace(mvo_hip_handle h, int32_t capacity);
int mvo_hip_get_trace(mvo_hip_handle h, mvo_hip_iteration *rows, int32_t capacity, int32_t *n_rows);

/* Device-side timing (HIP events around the launches) costs a few microseconds per solve and is
 * off by default; with it off mvo_hip_get_timing reports only the host-clock fields
 * (upload_seconds, solve_seconds), the launch counts and residual_blocks.  No reference
 * counterpart: the reference times refine() with ROS_INFO stamps around the call
 * (IterativeRefinement.cpp:231-234). */
int mvo_hip_set_profiling(mvo_hip_handle h, int enabled);
int mvo_hip_get_timing(mvo_hip_handle h, mvo_hip_timing *out);

/* The one entry point that allocates nothing.  NO COUNTERPART IN THE REFERENCE: it exists for
 * callers that cannot allocate -- code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc.  It is mvo_hip_refine for ONE window with every buffer in device memory:
 * frames_device[N] in/out (records [0, M) receive the refined vec and scale, as
 * IterativeRefinement.cpp:240-246), pair_offsets_device[P + 1], m0_device / m1_device
 * [3 * n_blocks], summary_device (may be NULL); options is host memory, read before the call
 * returns.  Three kernel launches are queued on `stream` (a hipStream_t; NULL = the null stream) of
 * device `device_ordinal`; the call does not wait for them, copies nothing and calls no hipMalloc,
 * hipFree or any other allocating routine.  All working memory (pair tables, per-pair sums, solver
 * state, result record) lives in one file-scope device array of MVO_HIP_NOALLOC_SCRATCH_BYTES that
 * every GPU the library runs on holds until the process ends.  Limits, rejected with
 * MVO_HIP_ERR_UNSUPPORTED: N <= MVO_HIP_MAX_FRAMES, M <= MVO_HIP_MAX_REFINE,
 * n_blocks <= MVO_HIP_NOALLOC_MAX_BLOCKS (the window is solved by one workgroup).  n_blocks must
 * equal pair_offsets_device[P]; that, and the offsets being non-decreasing from 0, is checked on
 * the device: on a mismatch no ray is read, the frames stay untouched and the summary reports
 * MVO_HIP_FAILURE / MVO_HIP_REASON_BAD_INPUT.  The initial t = reverseScale(scale) is evaluated with
 * the device's log instead of the host's (mvo_hip_refine), so results agree with mvo_hip_refine to
 * rounding, not bit for bit.  Every call on a GPU shares that GPU's array: calls that overlap on
 * one GPU (two streams, two threads) race; calls queued on one stream do not.  The first call on a
 * device resolves the array's address (hipGetSymbolAddress) -- make it outside graph capture. */
#define MVO_HIP_NOALLOC_SCRATCH_BYTES (192 * 1024)
#define MVO_HIP_NOALLOC_MAX_BLOCKS (1u << 22)
int mvo_hip_refine_noalloc(int device_ordinal, void *stream, mvo_hip_frame *frames_device, uint32_t M,
                           uint32_t N, const mvo_hip_options *options, const uint64_t *pair_offsets_device,
                           const double *m0_device, const double *m1_device, uint64_t n_blocks,
                           mvo_hip_summary *summary_device);

/* scalar helpers exported for bindings and tests (host code, libm):
 * reverseScale (IterativeRefinement.cpp:348-368) and scaleTemplated<double> (:298-309). */
double mvo_hip_reverse_scale(double length, double highest_length, double lowest_length);
double mvo_hip_scale(double t, double highest_length, double lowest_length);
uint32_t mvo_hip_num_pairs(uint32_t M, uint32_t N);
const char *mvo_hip_status_string(int status);
const char *mvo_hip_version(void);

/* Test aid (no counterpart in the reference): fills the LDS of every CU of `device_ordinal` with NaN
 * bit patterns and returns when that is done.  LDS is not cleared between kernels -- not even between
 * processes -- so a kernel that reads LDS it never wrote gives results that depend on what ran
 * before it; the GPU tests call this before every case (tests/conftest.py). */
int mvo_hip_debug_poison_lds(int device_ordinal);

/* Test aid: with MVO_HIP_POISON_ALLOC set in the environment every device allocation of the library
 * starts as 0xFF bytes and carries a 256-byte guard behind its end.  Returns the number of guards
 * found overwritten so far (at a free, or among the live allocations now; drains the device), or
 * -1 when the variable is not set. */
long mvo_hip_debug_check_guards(void);

#ifdef __cplusplus
}
#endif
#endif /* MVO_HIP_H */
