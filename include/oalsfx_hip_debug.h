/*
 * oalsfx_hip_debug.h -- measurement and test helpers of liboalsfx_hip.so: what bench.py, scripts/ and tests/ use to time kernels,
 * calibrate counters, probe memory placement and switch experiment paths.  Nothing here stands for a call of the reference, and a
 * caller of the effect path (include/oalsfx_hip.h) needs none of it.  Same conventions: 1 on success, 0 on failure.
 */
#ifndef OALSFX_HIP_DEBUG_H
#define OALSFX_HIP_DEBUG_H

#include "oalsfx_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* oalsfx_batch_mix with its three legs timed by HIP events on the batch's stream: legs_us[0] the copy in, [1] the kernels, [2] the copy
 * out, in microseconds (a measurement aid: bench.py's host_io object says with it where a slow box loses the time). */
int oalsfx_batch_mix_timed(oalsfx_batch* b, int frames, const float* src_host, float* dst_host, double legs_us[3]);
/* Measurement: of the instance hand-overs between chained calls so far, how many stayed on one CU (those pay for an L1 invalidate). Waits. */
long long oalsfx_debug_chain_same_cu(oalsfx_batch* b);
/* ---- synthetic input generator of the benchmark filled directly in device memory:
 * value(instance, k) for buffer `buffer_index`, identical to the oracle's generator. */
int oalsfx_batch_fill_synthetic(oalsfx_batch* b, int frames, unsigned buffer_index, float* dst_dev, void* hip_stream);

/* ---- HIP-event timing of the dominant kernel, measured on the launch stream.  enable = 0 switches it off, 1 times every
 * mix call, k > 1 every k-th (a timed launch costs a few microseconds of dispatch overhead).  The effect kernel launches of a
 * timed call are bracketed by a start / stop event pair; read() returns the number of launches of `effect_type`
 * since enable and their summed duration in milliseconds.  Every effect type of a slot other than the two reverbs shares
 * one launch (k_wave_effects), which any of those types reads.  The two reverb types share their launches too: either
 * type reads the steady-state kernel, type + 16 the general kernel (the groups of a slot run side by side).  32 reads the
 * grid that serves a slot's ring-light effects and steady reverbs together (k_slot_mixed: mono / stereo, whole tiles). */
int oalsfx_batch_kernel_timing(oalsfx_batch* b, int enable);
int oalsfx_batch_kernel_timing_read(oalsfx_batch* b, int effect_type, int* launches, double* total_ms);
/* The same launches one by one: up to `max_samples` durations in microseconds (event pair, uncorrected) into `out_us`; returns the
 * number of timed launches of that type since enable (-1 on error).  bench.py takes its median from these. */
int oalsfx_batch_kernel_timing_samples(oalsfx_batch* b, int effect_type, double* out_us, int max_samples);
/* What the placement search for the delay-line chunks did: chunks allocated, candidates probed, and the traffic-only probe's
 * microseconds per launch on the candidate kept last and on the slowest one seen next to it (0 when no search ran). */
int oalsfx_batch_placement(const oalsfx_batch* b, int* chunks, int* candidates, double* best_us, double* worst_us);
/* What such an event pair measures beyond the kernel: the average elapsed time of `repeats` pairs with nothing between them
 * on the batch's stream (about 4.4 us on MI355X).  bench.py reports its kernel time with this subtracted, which agrees with
 * rocprofv3's kernel trace of the same run to 2 %, and keeps the raw figure next to it. */
int oalsfx_batch_event_overhead(oalsfx_batch* b, int repeats, double* avg_us);

/* ---- measurement helper: sweeps a scratch buffer of `bytes` with the reverb kernel's access shape (one dword per lane,
 * 256 contiguous bytes per wave instruction), `repeats` launches of k_hbm_sweep, reading (write == 0) or writing.  Used under
 * rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE to calibrate those counters against a known byte count (profiles/README.md). */
int oalsfx_debug_hbm_sweep(int device_id, unsigned long long bytes, int write, int repeats);
/* ---- measurement helper: the experiment switches of OALSFX_DEBUG_FLAGS (hip/batch.cpp: debug_flags), settable between calls so
 * that one process can time two code paths side by side on the same box (scripts/ab_paths.py).  Process-wide. */
void oalsfx_debug_set_flags(int flags);
/* ---- measurement helper: device address of a slot's delay-line slab (0 if it has none): where a batch's slabs land in memory
 * moves the reverb kernel's launch time by a few per cent (scripts/placement_bench.py). */
unsigned long long oalsfx_debug_ring_address(oalsfx_batch* b, int instance, int slot);
/* ---- measurement helper: moves every delay-line chunk of the batch to a fresh allocation (contents copied, slab table updated);
 * keep_old != 0 leaves the old chunk allocated so that the next move lands elsewhere again.  Synchronises. */
int oalsfx_debug_move_rings(oalsfx_batch* b, int keep_old);
/* ---- measurement helper: k_stream_pattern (above) on the batch's own delay-line chunk, `repeats` launches, average microseconds per
 * launch.  Overwrites the delay lines: for placement experiments only. */
int oalsfx_debug_probe_rings(oalsfx_batch* b, int repeats, double* avg_us);
int oalsfx_debug_probe_pointer(void* slabs, int instances, int slab_floats, int repeats, double* avg_us);
/* ---- measurement helper: the ring traffic of the steady-state reverb kernel without its arithmetic (k_stream_pattern:
 * per instance 24 unaligned read streams and 24 aligned write streams of 256 frames per launch, `dwords_per_lane` = 1, 2 or 4
 * consecutive dwords per lane = 256-, 512- or 1024-byte bursts; slabs `slab_floats` apart (>= 235520), instance i shifted by
 * i * pos_skew samples inside its streams).  Returns the average launch time of `repeats` launches. */
int oalsfx_debug_stream_pattern(int device_id, int instances, int dwords_per_lane, int repeats, int slab_floats, int pos_skew, double* avg_us);

/* The pipelined host-pointer path (oalsfx_batch_mix_async): which form the batch settled on -- 3: copy in, kernels and copy out of
 * successive calls on three streams; 1: all three on the batch's one stream, in order; 0: still probing -- and what its probe saw per call
 * in either form (microseconds; 0 where no probe ran: the form was known for the device, or fixed by OALSFX_HOST_PIPELINE). */
int oalsfx_debug_host_pipeline(oalsfx_batch* b, int* form, double* probe_us_three_streams, double* probe_us_one_stream);

/* Test hook: every gate in front of a chained launch waits for `skew` more workgroups than will ever start, so that it gives up
 * (after its full wait, about a second) -- what a tool that runs kernels one at a time out of queue order does to it.  The results stay
 * whole; the batch notices at its next synchronising call and stays in stream order from then on (oalsfx_debug_chain_given_up). */
void oalsfx_debug_gate_skew(oalsfx_batch* b, unsigned skew);
int oalsfx_debug_chain_given_up(const oalsfx_batch* b);
/* Chained launches: the gate in front of a launch is set by the host's count of the workgroups started so far, which every
 * workgroup of a chained launch adds itself to on the device.  Reads both (waits for the batch): the two must agree after any run. */
int oalsfx_debug_chain_started(oalsfx_batch* b, unsigned* host_total, unsigned* device_total);

#ifdef __cplusplus
}
#endif

#endif /* OALSFX_HIP_DEBUG_H */
