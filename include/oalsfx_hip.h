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
/* The same for a caller whose instances each have a source and a target buffer of their own (n_instances pointers each, every buffer
 * frames * channels floats): what a program that held one oalsfxpp::Api object per voice has.  Gathered, mixed as one call, scattered. */
int oalsfx_batch_mix_gather(oalsfx_batch* b, int frames, const float* const* src_per_instance, float* const* dst_per_instance);
/* Same with buffers already resident in device memory; launches on `hip_stream` (a hipStream_t, or
 * NULL for the batch's own stream) and returns without synchronising.
 * With hip_stream NULL the call is complete when oalsfx_batch_synchronize (or any other call on the batch that waits or reads back)
 * returns; consecutive such calls may overlap on the device where the instances allow it (a step that is one steady-state reverb
 * launch, a launch of reverb-free slots followed by one of the reverbs' slot, or one grid of ring-light effects and proven reverbs:
 * each instance of the later call starts when the earlier call is through with that instance), which takes the launch gap
 * between dependent kernels out of a streaming loop.  A caller that wants the launches in the order of a stream it can queue its
 * own work on passes that stream, or asks for the batch's with oalsfx_batch_stream, which switches the overlap off. */
int oalsfx_batch_mix_device(oalsfx_batch* b, int frames, const float* src_dev, float* dst_dev, void* hip_stream);
int oalsfx_batch_synchronize(oalsfx_batch* b);
/* How many oalsfx_batch_mix_device calls overlapped with their neighbours that way so far (tests, benchmark records). */
long long oalsfx_batch_chained_calls(const oalsfx_batch* b);
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

/* ---- one instance range over several GPUs (BASELINE configs[4]: 262 144 EAX reverbs over the eight GPUs of a node).  The reference's
 * instances share nothing (src/oalsfxpp.cpp:2984-3037), so the split is a contiguous range per device -- sizes differ by at most one --,
 * one batch, one stream set and one host thread per device, no collective and no peer traffic.  device_ids lists the HIP ordinals (an
 * ordinal may appear more than once: two shards on one GPU, which is how the split is rehearsed on a one-GPU box).  Setters take ranges
 * of the global instance numbering; the buffers of oalsfx_group_mix hold the whole range, [n_total][frames][channels], in host memory,
 * and every device's thread copies its part in, runs its kernels and copies out (Api::mix, src/oalsfxpp.cpp:3785-3829, for every
 * instance); oalsfx_group_mix_device takes one device-resident source and target buffer per shard and only queues (consecutive calls
 * overlap on every device as for a batch alone), oalsfx_group_synchronize waits for all.  A failed call's message (oalsfx_group_error)
 * names the device; oalsfx_group_last_error is for a failed create.  Not thread-safe, like a batch. */
typedef struct oalsfx_group oalsfx_group;
oalsfx_group* oalsfx_group_create(int n_total, const int* device_ids, int n_devices, int channel_format, int sampling_rate, int effect_count);
void oalsfx_group_destroy(oalsfx_group* g);
const char* oalsfx_group_error(const oalsfx_group* g);
const char* oalsfx_group_last_error(void);
int oalsfx_group_instances(const oalsfx_group* g);
int oalsfx_group_channels(const oalsfx_group* g);
int oalsfx_group_devices(const oalsfx_group* g);
/* Shard k: its device ordinal, first global instance and instance count; and its batch, for what the group does not forward (read-backs,
 * oalsfx_batch_plan, the pipelined host path). */
int oalsfx_group_shard(const oalsfx_group* g, int k, int* device_id, int* first, int* count);
oalsfx_batch* oalsfx_group_batch(oalsfx_group* g, int k);
int oalsfx_group_set_effect(oalsfx_group* g, int first, int count, int slot, const oalsfx_effect* effects, int stride_bytes);
int oalsfx_group_set_effect_type(oalsfx_group* g, int first, int count, int slot, int effect_type);
int oalsfx_group_set_effect_props(oalsfx_group* g, int first, int count, int slot, const void* props, int stride_bytes);
int oalsfx_group_set_send_props(oalsfx_group* g, int first, int count, int slot, const oalsfx_send_props* props);
int oalsfx_group_apply_changes(oalsfx_group* g, int first, int count);
int oalsfx_group_mix(oalsfx_group* g, int frames, const float* src_host, float* dst_host);
int oalsfx_group_mix_device(oalsfx_group* g, int frames, const float* const* src_per_device, float* const* dst_per_device);
int oalsfx_group_synchronize(oalsfx_group* g);

/* ---- device memory the library keeps between batches.  Batches whose calls can overlap on the device keep their delay lines, effect
 * state and hot records in uncached device memory, which the library takes from the runtime in 2 MiB granules and keeps for the next
 * batch that asks for a block of that size -- by default for the life of the process: on this runtime, uncached blocks given back with
 * hipFree have been seen to disturb ordinary allocations made afterwards (profiles/r04b_uncached_free_hazard/).  A process
 * that needs the memory back calls oalsfx_trim_pools: it waits for the device, frees everything that waits for reuse and returns the
 * bytes freed; OALSFX_UNCACHED_POOL_MAX_GIB=<GiB> (environment) does the same to whatever exceeds that much whenever a batch is
 * destroyed.  oalsfx_pools_waiting_bytes says how much waits.  Process-wide. */
unsigned long long oalsfx_trim_pools(void);
unsigned long long oalsfx_pools_waiting_bytes(void);

/* ---- state read-back for tests and checkpoints (the reference keeps this in private members of
 * the EffectState subclasses). */
int oalsfx_batch_read_slot(oalsfx_batch* b, int instance, int slot, oalsfx_slot_params* params, oalsfx_slot_state* state);
/* Copies up to max_floats of the slot's delay rings; returns the ring size in floats (0: no ring). */
int oalsfx_batch_read_ring(oalsfx_batch* b, int instance, int slot, float* out, int max_floats);
int oalsfx_batch_read_source(oalsfx_batch* b, int instance, oalsfx_source_params* params, oalsfx_source_state* state);

/* How the next mix call would lay out `slot` (pending property changes and read-backs folded in first): counts[0] instances on the
 * ring-light kernels, [1] reverbs proven steady (the builds without fallback), [2] reverbs believed steady, [3] reverbs on
 * the general kernel.  Nothing the reference has a counterpart for; tests and bench.py use it to say which kernel they measured. */
int oalsfx_batch_plan(oalsfx_batch* b, int slot, int counts[4]);
/* Symbol of the steady-state reverb kernel launched last, with its template arguments as rocprofv3 prints them ("" before the first). */
const char* oalsfx_batch_last_reverb_kernel(const oalsfx_batch* b);
/* PCI bus id ("0000:c1:00.0") of a HIP device ordinal, for benchmark records that must show N distinct GPUs.  Returns 1 on success. */
int oalsfx_device_pci_bus_id(int device_id, char* out, int len);

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

/* Measurement and test helpers of bench.py, scripts/ and tests/ (kernel timing by events, counter calibration, placement probes, the
 * experiment switches): include/oalsfx_hip_debug.h.  Exported by the same library; nothing a caller of the effect path needs. */

#endif /* OALSFX_HIP_H */
