/*
 * oalsfx_hip.h -- C ABI of liboalsfx_hip.so, the MI355X batch backend for the
 * oalsfxpp effect-process hot path.
 *
 * The reference has no FFI: its only seam is C++ (class oalsfxpp::Api, reference
 * src/oalsfxpp.h:760-922, and the internal EffectState::process virtual,
 * src/oalsfxpp.cpp:2239-2246).  This ABI is what a binding of that seam needs when
 * thousands of independent Api instances are advanced together: each call below
 * names the reference entry point it stands in for.  All handles are opaque, all
 * buffers are caller-owned plain pointers, no C++ or torch types cross the boundary.
 *
 * Conventions (same as the reference): calls return 1 on success and 0 on
 * failure; oalsfx_batch_error() then returns a static message.  Samples are
 * interleaved fp32 frames, [instance][frame][channel], and outputs are not clipped.
 * A batch is not thread-safe; distinct batches are independent.
 */
#ifndef OALSFX_HIP_H
#define OALSFX_HIP_H

#include "oalsfx_desc.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oalsfx_batch oalsfx_batch;

/* Mirror of oalsfxpp::Effect (reference src/oalsfxpp.h:532-548): 4-byte type tag followed by
 * the 108-byte EffectProps union.  sizeof == 112. */
typedef struct {
    int32_t type;
    unsigned char props[108];
} oalsfx_effect;

/* Mirror of oalsfxpp::SendProps (reference src/oalsfxpp.h:550-581). */
typedef struct { float gain, gain_hf, gain_lf; } oalsfx_send_props;

/* ---- lifecycle: n_instances x Api::initialize (reference src/oalsfxpp.cpp:3480-3504, 2846-2905).
 * channel_format is an OALSFX_FMT_* value; device_id is the HIP device ordinal.  Fails (returns
 * NULL and sets the global message readable through oalsfx_last_error) when the arguments are
 * out of range or no HIP device is usable -- there is no CPU fallback. */
oalsfx_batch* oalsfx_batch_create(int n_instances, int channel_format, int sampling_rate, int effect_count, int device_id);
void oalsfx_batch_destroy(oalsfx_batch* b);                     /* Api::uninitialize, src/oalsfxpp.cpp:3831 */
const char* oalsfx_batch_error(const oalsfx_batch* b);          /* Api::get_error_message, src/oalsfxpp.cpp:3836 */
const char* oalsfx_last_error(void);                            /* message of a failed oalsfx_batch_create */

int oalsfx_batch_instances(const oalsfx_batch* b);
int oalsfx_batch_channels(const oalsfx_batch* b);               /* Api::get_channel_count, src/oalsfxpp.cpp:3533 */
int oalsfx_batch_sampling_rate(const oalsfx_batch* b);          /* Api::get_sampling_rate, src/oalsfxpp.cpp:3511 */
int oalsfx_batch_effect_count(const oalsfx_batch* b);           /* Api::get_effect_count, src/oalsfxpp.cpp:3544 */

/* ---- deferred property setters for the instance range [first, first+count).
 * `stride_bytes` is the distance between consecutive per-instance records at `effects` /
 * `props`; 0 broadcasts one record to the whole range. */
/* Api::set_effect (src/oalsfxpp.cpp:3639-3658; this ABI reports success as 1, the C++ facade keeps
 * the reference's quirk of returning false). */
int oalsfx_batch_set_effect(oalsfx_batch* b, int first, int count, int slot, const oalsfx_effect* effects, int stride_bytes);
/* The same for instances that are not neighbours: instances[k] gets effects[k] (stride_bytes apart; 0: one effect for all).  One call
 * where a loop over oalsfx_batch_set_effect would make one per instance -- a foreign-function call costs more than the setter. */
int oalsfx_batch_set_effect_at(oalsfx_batch* b, const int* instances, int count, int slot, const oalsfx_effect* effects, int stride_bytes);
/* Api::set_effect_type (src/oalsfxpp.cpp:3597-3616): type tag + that type's default properties. */
int oalsfx_batch_set_effect_type(oalsfx_batch* b, int first, int count, int slot, int effect_type);
/* Api::set_effect_props (src/oalsfxpp.cpp:3618-3637): replaces the 108-byte union only. */
int oalsfx_batch_set_effect_props(oalsfx_batch* b, int first, int count, int slot, const void* props, int stride_bytes);
/* Api::set_send_props (src/oalsfxpp.cpp:3712-3736); slot < 0 addresses the direct send. */
int oalsfx_batch_set_send_props(oalsfx_batch* b, int first, int count, int slot, const oalsfx_send_props* props);
/* Api::get_effect / get_deferred_effect (src/oalsfxpp.cpp:3555-3595) for one instance. */
int oalsfx_batch_get_effect(const oalsfx_batch* b, int instance, int slot, int deferred, oalsfx_effect* out);
/* Api::get_send_props / get_deferred_send_props (src/oalsfxpp.cpp:3660-3710). */
int oalsfx_batch_get_send_props(const oalsfx_batch* b, int instance, int slot, int deferred, oalsfx_send_props* out);
/* Api::apply_changes (src/oalsfxpp.cpp:3738-3783) on every instance of the range. */
int oalsfx_batch_apply_changes(oalsfx_batch* b, int first, int count);

/* ---- the hot path: Api::mix (src/oalsfxpp.cpp:3785-3829) for every instance at once.
 * src and dst hold n_instances * frames * channels floats.  frames may be any positive count;
 * more than 2048 are processed in 2048-frame chunks like the reference.  frames == 0 succeeds. */
int oalsfx_batch_mix(oalsfx_batch* b, int frames, const float* src_host, float* dst_host);
/* The same call with its three legs timed by HIP events on the batch's stream: legs_us[0] the copy in, [1] the kernels, [2] the copy
 * out, in microseconds (a measurement aid: bench.py's host_io object says with it where a slow box loses the time). */
int oalsfx_batch_mix_timed(oalsfx_batch* b, int frames, const float* src_host, float* dst_host, double legs_us[3]);
/* Same with buffers already resident in device memory; launches on `hip_stream` (a hipStream_t, or
 * NULL for the batch's own stream) and returns without synchronising.
 * With hip_stream NULL the call is complete when oalsfx_batch_synchronize (or any other call on the batch that waits or reads back)
 * returns; consecutive such calls may overlap on the device where the instances allow it (a step that is one steady-state reverb
 * launch: each instance of the later call starts when the earlier call is through with that instance), which takes the launch gap
 * between dependent kernels out of a streaming loop.  A caller that wants the launches in the order of a stream it can queue its
 * own work on passes that stream, or asks for the batch's with oalsfx_batch_stream, which switches the overlap off. */
int oalsfx_batch_mix_device(oalsfx_batch* b, int frames, const float* src_dev, float* dst_dev, void* hip_stream);
int oalsfx_batch_synchronize(oalsfx_batch* b);
/* How many oalsfx_batch_mix_device calls overlapped with their neighbours that way so far (tests, benchmark records). */
long long oalsfx_batch_chained_calls(const oalsfx_batch* b);
/* Measurement: of the instance hand-overs between chained calls so far, how many stayed on one CU (those pay for an L1 invalidate). Waits. */
long long oalsfx_debug_chain_same_cu(oalsfx_batch* b);
/* Api::mix for a caller that streams buffer after buffer from host memory (what the reference's only entry point is used for,
 * src/oalsfxpp.cpp:3785-3829, src/oalsfxpp_test.cpp:891): returns as soon as the call is queued.  The copy in of call k + 1, the
 * kernels of call k and the copy out of call k - 1 overlap on three streams, ordered by events.  `src_host` and `dst_host` must stay
 * untouched until the call is through, which is the case once oalsfx_batch_wait has returned, or once three further
 * oalsfx_batch_mix_async calls have (a call first waits for the one that used its staging slot three calls earlier).  The copies
 * only run beside the kernels when the host buffers are page-locked (oalsfx_pinned_alloc, or the caller's own hipHostMalloc /
 * hipHostRegister); pageable buffers work but serialise.  Property setters and apply_changes may be called between the calls as
 * with oalsfx_batch_mix; they take effect from the next call on. */
int oalsfx_batch_mix_async(oalsfx_batch* b, int frames, const float* src_host, float* dst_host);
/* Waits until every queued oalsfx_batch_mix_async call has delivered its output. */
int oalsfx_batch_wait(oalsfx_batch* b);
/* Page-locked host memory for the two calls above (hipHostMalloc / hipHostFree), so that a caller need not link HIP itself. */
void* oalsfx_pinned_alloc(unsigned long long bytes);
void oalsfx_pinned_free(void* p);
/* The batch's own HIP stream (a hipStream_t) so callers can bracket launches with their own events.  From this call on every launch of
 * oalsfx_batch_mix_device(..., NULL) is in the order of that stream (no overlap of consecutive calls). */
void* oalsfx_batch_stream(oalsfx_batch* b);

/* ---- state read-back for tests and checkpoints (the reference keeps this in private members of
 * the EffectState subclasses). */
int oalsfx_batch_read_slot(oalsfx_batch* b, int instance, int slot, oalsfx_slot_params* params, oalsfx_slot_state* state);
/* Copies up to max_floats of the slot's delay rings; returns the ring size in floats (0: no ring). */
int oalsfx_batch_read_ring(oalsfx_batch* b, int instance, int slot, float* out, int max_floats);
int oalsfx_batch_read_source(oalsfx_batch* b, int instance, oalsfx_source_params* params, oalsfx_source_state* state);

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
/* How the next mix call would lay out `slot` (pending property changes and read-backs folded in first): counts[0] instances on the
 * ring-light kernels, [1] reverbs proven steady (the builds without fallback), [2] reverbs believed steady, [3] reverbs on
 * the general kernel.  Nothing the reference has a counterpart for; tests and bench.py use it to say which kernel they measured. */
int oalsfx_batch_plan(oalsfx_batch* b, int slot, int counts[4]);
/* What the placement search for the delay-line chunks did: chunks allocated, candidates probed, and the traffic-only probe's
 * microseconds per launch on the candidate kept last and on the slowest one seen next to it (0 when no search ran). */
int oalsfx_batch_placement(const oalsfx_batch* b, int* chunks, int* candidates, double* best_us, double* worst_us);
/* Symbol of the steady-state reverb kernel launched last, with its template arguments as rocprofv3 prints them ("" before the first). */
const char* oalsfx_batch_last_reverb_kernel(const oalsfx_batch* b);
/* PCI bus id ("0000:c1:00.0") of a HIP device ordinal, for benchmark records that must show N distinct GPUs.  Returns 1 on success. */
int oalsfx_device_pci_bus_id(int device_id, char* out, int len);
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

/* ---- host-only helpers (no GPU needed): the parameter-update path, exposed so the descriptors can be
 * checked against the reference and so the CPU oracle can be driven with identical parameters. */
void oalsfx_host_effect_defaults(int effect_type, oalsfx_effect* out);        /* Effect::set_type_and_defaults */
void oalsfx_host_effect_normalize(oalsfx_effect* e);                          /* Effect::normalize */
int oalsfx_host_derive_slot(int channel_format, int sampling_rate, const oalsfx_effect* normalized, oalsfx_slot_params* out);
int oalsfx_host_derive_source(int channel_format, int sampling_rate, int effect_count, const oalsfx_send_props* direct,
                              const oalsfx_send_props* aux /* [effect_count] */, const int* slot_types /* [effect_count] */,
                              oalsfx_source_params* out);
int oalsfx_host_ring_floats(int effect_type, int sampling_rate);
int oalsfx_host_channel_count(int channel_format);
int oalsfx_host_preset_count(void);
const char* oalsfx_host_preset_name(int index);
int oalsfx_host_preset(int index, void* reverb_props_out /* 108 bytes */);

#ifdef __cplusplus
}
#endif

#endif /* OALSFX_HIP_H */
