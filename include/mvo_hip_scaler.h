/* mvo_hip_scaler.h -- C ABI of the closed-form scale stage.
 *
 * Replaces the arithmetic of Scaler::stage (/root/reference/mvo/src/pipeline/Scaler.cpp:13-69), the
 * pipeline stage right before the refiner: it produces the initial `scale` of the newest frame
 * that IterativeRefinement::refine then starts from.  The frame bookkeeping (_prevFrame /
 * _keptFrame hand-over, Scaler.cpp:76-79) and the 3-frame correspondence gather
 * (Frame::getCorrespondingFeatures, sliding_window/Frame.hpp:102-143) stay on the caller's side.
 *
 * Per track through the 3-frame window (Scaler.cpp:34-44), rays as the reference holds them
 * (x, y, 1), any norm:
 *     m_k = (R_k * v_k) / |v_k|,  k = 0, 1, 2
 *     n0  = n1 * sqrt(1 - (m2.u1)^2) * sqrt(1 - (m1.m0)^2)
 *              / ( sqrt(1 - (m2.m1)^2) * sqrt(1 - (u1.m0)^2) )
 * then the median of the n0 (Scaler.cpp:48-55: sorted ascending, even count = mean of the two
 * middle values) clamped to [lowest_length, highest_length] (Scaler.cpp:57-67).
 * Inputs are taken as the reference passes them, including its pairing of rotations and rays
 * (R2 = kept frame, R1 = previous frame, Scaler.cpp:18-19, while v1 / v2 come from the kept /
 * previous frame, Scaler.cpp:22): the library applies the formula, it does not second-guess it.
 * NaN scales (a negative radicand) are ordered last; the reference sorts them with std::sort,
 * whose result is then unspecified.
 */
#ifndef MVO_HIP_SCALER_H
#define MVO_HIP_SCALER_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct mvo_hip_scaler_context *mvo_hip_scaler_handle;

typedef struct {
  double R0[9], R1[9], R2[9]; /* row-major, Scaler.cpp:18-20 */
  double u1[3];               /* _keptFrame->getBaseLineToPrevious(), Scaler.cpp:17 */
  double n1;                  /* _keptFrame->getScaleToPrevious(), Scaler.cpp:16 */
  double lowest_length, highest_length; /* params.lowestLength / highestLength, Scaler.cpp:57-63 */
} mvo_hip_scaler_input; /* 264 bytes */

typedef struct {
  double scale;      /* what Scaler::stage hands to setScaleToPrevious */
  double median;     /* before clamping */
  uint64_t n_tracks; /* 0: no correspondence through the window (Scaler.cpp:25-26), scale untouched */
  int32_t clamped;   /* -1 raised to lowest_length, +1 cut to highest_length, 0 neither */
  int32_t reserved;
} mvo_hip_scaler_result; /* 32 bytes */

/* status codes are those of mvo_hip.h (0 = ok) */
int mvo_hip_scaler_create(int device_ordinal, mvo_hip_scaler_handle *out);
void mvo_hip_scaler_destroy(mvo_hip_scaler_handle h);

/* one call = lines 21-68 of Scaler::stage; v0, v1, v2: 3 doubles per track, host memory;
 * per_track (n doubles, host) may be NULL */
int mvo_hip_scaler_scale(mvo_hip_scaler_handle h, const mvo_hip_scaler_input *in, uint64_t n_tracks,
                         const double *v0, const double *v1, const double *v2, double *per_track,
                         mvo_hip_scaler_result *result);

/* split form: rays stay resident in HBM (bench, replay) */
int mvo_hip_scaler_upload(mvo_hip_scaler_handle h, uint64_t n_tracks, const double *v0, const double *v1,
                          const double *v2);
int mvo_hip_scaler_run(mvo_hip_scaler_handle h, const mvo_hip_scaler_input *in, double *per_track,
                       mvo_hip_scaler_result *result);
/* seconds the device spent in the last run (HIP events), and its algorithmic bytes: 72 B of rays in,
 * 8 B of scale out, and for more than 2048 tracks one more pass over the 8-byte scales for the median
 * (two when the first 64 tracks are not representative of the rest) */
int mvo_hip_scaler_last_timing(mvo_hip_scaler_handle h, double *device_seconds, uint64_t *bytes);
/* the same for the dominant kernel of that run alone, k_scaler_tracks (80 B per track); zeros for a
 * run of at most 2048 tracks, which is one fused launch */
int mvo_hip_scaler_tracks_timing(mvo_hip_scaler_handle h, double *device_seconds, uint64_t *bytes);

#ifdef __cplusplus
}
#endif
#endif /* MVO_HIP_SCALER_H */
