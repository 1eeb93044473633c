// mvo_internal.h -- entry points shared between the translation units of libmvo_hip.so only (not
// part of the public C ABI in include/).
#pragma once
#include "../../include/mvo_hip.h"
#include "../../include/mvo_hip_scaler.h"

extern "C" {
void* mvo_hip_internal_stream(mvo_hip_handle h);  // hipStream_t of the handle
int mvo_hip_internal_device(mvo_hip_handle h);
// mvo_hip_upload for ONE window whose rays the caller writes on the device: observations->m0 / m1 are
// not read (non-NULL placeholders), pair_offsets is host memory.  Returns where the 3-double rays of
// residual block k go (m0_device + 3 k); the caller fills them with work queued on the handle's
// stream before mvo_hip_solve.
int mvo_hip_internal_upload_device_rays(mvo_hip_handle h, const mvo_hip_frame* frames, uint32_t M, uint32_t N,
                                        const mvo_hip_options* options, const mvo_hip_observations* observations,
                                        double** m0_device, double** m1_device);
// The same for ONE window whose per-pair second moments the caller forms on the device (no loss): no ray
// buffers at all.  pair_offsets (host) gives the pairs' block counts.  Returns the device copy of the N
// frame records (valid once the work queued on the handle's stream has run) and where the 8 sums of pair
// p go (pair_sums_device + 8 p: S xx,xy,xz,yy,yz,zz, two unused); the caller fills them with work queued
// on the handle's stream before mvo_hip_solve, which then runs the LM loop only.
int mvo_hip_internal_upload_pair_sums(mvo_hip_handle h, const mvo_hip_frame* frames, uint32_t M, uint32_t N,
                                      const mvo_hip_options* options, const mvo_hip_observations* observations,
                                      const double** frames_device, double** pair_sums_device);
// scale stage: room for n_tracks tracks in the handle's own ray buffers (3 doubles per track each), which
// the caller then fills on the device; the next mvo_hip_scaler_run works on them.  *device is the
// handle's GPU.
int mvo_hip_scaler_internal_reserve(mvo_hip_scaler_handle h, uint64_t n_tracks, double** v0, double** v1, double** v2,
                                    int* device);
}
