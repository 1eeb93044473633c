/* mvo_hip_tracks.h -- device-resident feature-track tables and on-device observation gathering
 * (SURVEY.md section 8f, rank 2).
 *
 * The reference keeps, per keyframe, a vector of Feature {_positionProjected, _preFeature,
 * _preFeatureCounter} (/root/reference/mvo/src/sliding_window/Feature.hpp:6-22) chained frame to
 * frame, and IterativeRefinement::refine walks those chains on the host for every frame pair
 * (IterativeRefinement.cpp:116-165 through Frame::getCorrespondingFeatures,
 * sliding_window/Frame.hpp:258-302, and FeatureOperations::normFeatures,
 * operations/FeatureOperations.cpp:117-122).  Here the same table lives in HBM: each new keyframe
 * is pushed ONCE (its rays and predecessor indices); a refine call gathers the residual blocks of
 * all frame pairs on the device, straight into the solver's buffers, and uploads only the N frame
 * records.  _preFeatureCounter is derived on the device exactly as
 * Frame::calculateFeaturePreCounter does (sliding_window/Frame.cpp:132-148).
 */
#ifndef MVO_HIP_TRACKS_H
#define MVO_HIP_TRACKS_H
#include <stdint.h>

#include "mvo_hip.h"
#include "mvo_hip_scaler.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mvo_hip_tracks_context *mvo_hip_tracks_handle;

/* The table is a ring of max_frames keyframes with at most max_features features each, bound to
 * `solver` (its device and stream; the solver must outlive the table).  HBM per cell: 24 B unit ray,
 * 4 B predecessor, 4 B counter and 16 B of skip links (the track's feature 16 / 32 / 64 / 128 keyframes
 * back, derived at push; they let the gather start anywhere along a track after at most four lookups
 * instead of walking up to N - 1 dependent links): 48 B x max_frames x max_features. */
int mvo_hip_tracks_create(mvo_hip_handle solver, uint32_t max_frames, uint32_t max_features,
                          mvo_hip_tracks_handle *out);
void mvo_hip_tracks_destroy(mvo_hip_tracks_handle t);

/* Appends the newest keyframe.  rays: 3 doubles per feature as Feature::_positionProjected holds
 * them (any norm); pre_feature: Feature::_preFeature, the index of the same track in the
 * previously pushed frame, or negative.  The first frame ever pushed has no predecessor. */
int mvo_hip_tracks_push(mvo_hip_tracks_handle t, uint32_t n_features, const double *rays, const int32_t *pre_feature);

/* Appends features WITHOUT predecessor to the newest pushed keyframe, behind its existing ones: what
 * Frame::mergeFrame does to the previous keyframe when a later image is merged onto it
 * (sliding_window/Frame.cpp:103-119, called from pipeline/Merger.cpp:105-110).  Features of the
 * next pushed keyframe may name them as predecessors. */
int mvo_hip_tracks_append(mvo_hip_tracks_handle t, uint32_t n_features, const double *rays);

/* Forget everything (a new sequence). */
int mvo_hip_tracks_reset(mvo_hip_tracks_handle t);

/* IterativeRefinement::refine on the newest N pushed keyframes: `frames` as in mvo_hip_refine
 * (index 0 = newest), observations gathered on the device.  Same status codes, same results as
 * mvo_hip_refine on the host-gathered observations. */
int mvo_hip_tracks_refine(mvo_hip_tracks_handle t, mvo_hip_frame *frames, uint32_t M, uint32_t N,
                          const mvo_hip_options *options, mvo_hip_summary *summary);

/* IterativeRefinement::refine on the newest N pushed keyframes ... without a loss function the library
 * does not gather the residual blocks at all: the LM loop needs 6 second moments per frame pair
 * (DESIGN.md section 2), and those are summed straight from the track tables (k_tracks_fold).  parity/debug:
 * these moments, S_p = sum c c^T with c = (R_a m0) x (R_{b+1} m1) over the pair's blocks
 * (IterativeRefinement.cpp:116-172, 252-263), 6 doubles per pair (xx, xy, xz, yy, yz, zz) in the
 * reference's pair order -- what mvo_hip_pair_moments returns for the host-gathered blocks, to
 * rounding (the summation order differs). */
int mvo_hip_tracks_pair_moments(mvo_hip_tracks_handle t, const mvo_hip_frame *frames, uint32_t M, uint32_t N,
                                double *moments);

/* parity/debug: the gathered residual blocks of the newest N frames in the reference's pair order.
 * pair_offsets: n_pairs + 1 entries; m0 / m1 (3 doubles per block, capacity `capacity` blocks) may
 * be NULL to query the total = pair_offsets[n_pairs] first. */
int mvo_hip_tracks_gather(mvo_hip_tracks_handle t, uint32_t M, uint32_t N, uint64_t *pair_offsets, double *m0,
                          double *m1, uint64_t capacity);

/* The scale stage on the same tables: Scaler::stage's three-frame gather
 * (pipeline/Scaler.cpp:22 -> Frame::getCorrespondingFeatures(oldest, newest, {v0, v1, v2}),
 * sliding_window/Frame.hpp:103-152) runs on the device -- the features of the keyframe `age` pushes
 * ago (0 = newest) whose track reaches back two more keyframes, in that keyframe's feature order;
 * v0 = their rays in it, v1 / v2 = one / two keyframes older -- and feeds mvo_hip_scaler_run
 * (include/mvo_hip_scaler.h: lines 21-68 of the stage) without the rays leaving HBM.  Both handles
 * must live on the same GPU.  per_track may be NULL.  MVO_HIP_ERR_STATE with fewer than age + 3
 * keyframes in the table.  The rays reach the formula already unit-normalised (the table stores them so);
 * Scaler.cpp:35-37 divides by the norm once more, which changes nothing but the last bit. */
int mvo_hip_tracks_scale(mvo_hip_tracks_handle t, mvo_hip_scaler_handle scaler, uint32_t age,
                         const mvo_hip_scaler_input *in, double *per_track, mvo_hip_scaler_result *result);

/* Measurement (off by default): HIP events on the solver's stream around the two gather kernels of
 * the last refine / gather call. */
typedef struct {
  double count_seconds;      /* k_tracks_count + k_tracks_scan; 0 after a fused refine (no loss): it counts
                              * nothing and waits for nothing before the solve is queued */
  double fill_seconds;       /* k_tracks_fill, or k_tracks_fold when the refine ran fused */
  uint64_t residual_blocks;  /* gathered by the last call; after a fused refine they are counted when this
                              * record is asked for (once per table state and window shape) */
  uint64_t fill_bytes;       /* algorithmic bytes of that k_tracks_fill launch: 32 B per feature of the M
                              * newest frames (counter, predecessor, ray); per gathered block 4 B of
                              * predecessor index, 24 B of ray read, 2 x 24 B written (fused fold: nothing
                              * written per block, 28 B per block) */
} mvo_hip_tracks_timing; /* 32 bytes */
int mvo_hip_tracks_set_profiling(mvo_hip_tracks_handle t, int enabled);
int mvo_hip_tracks_get_timing(mvo_hip_tracks_handle t, mvo_hip_tracks_timing *out);

#ifdef __cplusplus
}
#endif
#endif /* MVO_HIP_TRACKS_H */
