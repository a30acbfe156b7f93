// Batch runtime behind the C ABI of include/oalsfx_hip.h.
//
// Owns, for N independent instances: the API-visible property bookkeeping (host), the derived
// descriptors (host shadow + device copy), the process-path state and delay rings (device only) and
// the launch plan (per slot, per effect type, a device list of instance indices).  One call to
// oalsfx_batch_mix* advances every instance by one buffer: the slot loop of the reference's
// Api::Impl::mix_data (reference src/oalsfxpp.cpp:2984-3037) becomes, per slot, one kernel launch per
// effect type present in that slot.
//
// There is no CPU implementation of the sample path in this library: without a usable HIP device
// oalsfx_batch_create fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../host/core.hpp"
#include "common.hpp"
#include "oalsfx_hip.h"
#include "oalsfx_hip_debug.h"

using namespace oalsfx_host;
using oalsfx_hip::KernelCtx;

namespace {

thread_local std::string g_last_error;

struct RingPool {
    size_t slab_floats = 0;
    std::vector<float*> free_clean; // zero-filled, never used since allocation
    std::vector<float*> free_dirty;
};

struct TimedLaunch { hipEvent_t start, stop; int type; };

constexpr int kSideStreams = 3; // ring-light effects, proven-steady reverbs, believed-steady reverbs, general reverbs: four kernel groups side by side at most

} // namespace

// How many launches of a run of chained launches may be in flight (streams taken in turn).  Two overlap one launch's tail with the next
// one's head; a third keeps the workgroup slots that fast workgroups free in use while the slowest of the launch two before are still
// at work (cross-fading instances, presets of the slower kinds): a launch starts when the launch kChainDepth before it has completed.
#ifndef OALSFX_CHAIN_DEPTH
#define OALSFX_CHAIN_DEPTH 3
#endif
constexpr int kChainDepth = OALSFX_CHAIN_DEPTH;
static_assert(kChainDepth >= 1 && kChainDepth <= 3, "a wavefront looks at the CUs of the two launches before it (reverb.hip, turn_cu / turn_cu2): at most three launches in flight");

struct oalsfx_batch {
    int n = 0, slots = 0, channels = 0, rate = 0, device = 0;
    int format = 0;
    DeviceDesc dev;
    std::vector<InstanceHost> inst;

    // host shadows
    std::vector<oalsfx_slot_params> h_params;     // [n*slots]
    std::vector<oalsfx_slot_state> h_state_init;  // [n*slots] staging for restarted slots
    std::vector<oalsfx_source_params> h_source;   // [n]
    std::vector<uint32_t> seq;                    // [n*slots]
    std::vector<float*> h_rings;                  // [n*slots]
    std::vector<size_t> ring_floats;              // [n*slots] size class of the slab held
    std::vector<uint8_t> inst_dirty;              // [n]
    // Host-side belief about the reverb slots, kept incrementally (a parameter change touches its own slot only):
    std::vector<int> since_update;                // [n*slots] frames mixed since the slot's last parameter update (capped)
    std::vector<uint8_t> slot_class;              // [n*slots] kClass* bits of the slot's current parameters
    std::vector<size_t> settling;                 // reverb slots updated less than kSettleFrames ago
    std::vector<uint8_t> in_settling;             // [n*slots] membership flag of `settling`
    std::vector<uint8_t> mod_ever;                // [n*slots] the slot's late line was modulated at some time since its state was created (the depth
                                                  // smoother keeps moving long after the depth is set to 0)
    std::vector<uint8_t> xf_ok;                   // [n*slots] settling, and in a way the cross-fading build of the steady-state kernel follows
                                                  // (taps of both sets a tile away, state kept): listed with the believed-steady instances
    int n_close[OALSFX_MAX_SLOTS] = {};           // per slot: reverbs with a tap between one and two tiles ...
    int n_short[OALSFX_MAX_SLOTS] = {};           // ... with a tap shorter than one tile
    bool modulated[OALSFX_MAX_SLOTS] = {};        // some reverb of the slot has, or had, a modulated late line (sticky: the depth
                                                  // smoother keeps moving long after the depth is set to 0)
    int n_filtered = 0;                           // instances with a send filter switched on
    std::vector<uint8_t> inst_filtered;           // [n] ... which
    std::vector<int> h_lists;                     // host copy of d_lists as last built
    uint32_t lists_version = 0;
    // how many filtered instances the SF builds take care of themselves (launch_reverb_kinds_part), cached per list version and boundary
    uint32_t inside_version = ~0u;
    int inside_end = -1, inside_filtered = 0;
    // What the host *knows* about the reverb slots (as opposed to believes): a slot is proven steady once the device has reported it
    // settled and at rest (d_exact, read back without ever waiting for the stream) and nothing has been uploaded for its instance since.
    std::vector<uint8_t> proven;                  // [n*slots] 0, or from how many tiles a block on the slot's output gains are at rest (1 .. 4)
    std::vector<uint32_t> updated_gen;            // [n*slots] upload generation of the last parameter upload that touched the slot's instance
    std::vector<uint32_t> inst_epoch;             // [n] stamps the hot records: bumped with every upload that touches the instance
    uint32_t upload_gen = 0;
    bool exact_wanted = false;                    // something changed that may let the next read-back prove more slots
    bool exact_pending = false;                   // a read-back of d_exact is in flight (ev_exact)
    uint32_t exact_gen = 0;                       // upload generation the read-back in flight was taken at
    std::vector<int> dirty_list;
    bool lists_dirty = true;
    // Instances a setter has written since they were last applied: apply_changes() visits only these (for the others the
    // reference's comparison of deferred and active properties finds nothing by construction).
    std::vector<uint8_t> touched;                 // [n]
    std::vector<int> touched_list;

    // device
    oalsfx_slot_params* d_params = nullptr;
    oalsfx_hip::SlotStateLines* d_state = nullptr;
    oalsfx_source_params* d_source = nullptr;
    float** d_rings = nullptr;
    oalsfx_source_state* d_source_state = nullptr; // [n] histories of the send filters
    float* d_filtered = nullptr;                  // [1 + slots][n][chunk frames][channels] outputs of the send-filter pre-pass
    size_t filtered_capacity = 0;                 // floats per send plane
    float* d_mixbuf = nullptr;
    int* d_lists = nullptr;                       // [slots][n]: the one of d_lists_buf the next launches read
    int* d_lists_buf[kChainDepth + 1] = {};       // every rebuilt list goes to the next buffer: a launch that is still in flight -- chained
                                                  // launches -- keeps reading the list it was given
    int lists_turn = 0;
    int* d_progress = nullptr;                    // [n*slots] hand-off from the steady-state reverb kernel to the general kernel behind it
    unsigned* d_hot = nullptr;                    // [n*slots][hot::SIZE] start records of the proven-steady reverb kernel
    unsigned* d_inst_epoch = nullptr;             // [n]
    unsigned* d_exact = nullptr;                  // [n*slots] "settled and at rest" as the reverb kernels left it
    unsigned* h_exact = nullptr;                  // pinned copy of d_exact, filled by the read-back
    char fault_text[400] = {};
    long long host_prepare_ns = 0, host_stage_wait_ns = 0, host_derive_ns = 0, host_lists_ns = 0; // OALSFX_HOST_PROFILE: where the host's time inside mix_device goes
    unsigned* h_fault = nullptr;                  // pinned, device-visible: instances a proven-steady launch had to leave alone (must stay 0)
    unsigned* d_fault = nullptr;                  // device address of h_fault  ([1]: gates of chained launches that gave up waiting, a word of its own)
    bool chain_given_up = false;                  // gates counted out and nothing else did: something runs the queues' kernels one at a time; stream order from then on
    uint32_t gate_skew = 0;                       // test hook (oalsfx_debug_gate_skew): added to every gate's target, so that the gates count out
    hipEvent_t ev_exact = nullptr;
    // Per slot the list is: ring-light types in ascending order (list_offset / list_count per type), then the reverb instances
    // proven steady (reverb, EAX reverb: steady_offset / fast_count), those believed steady (reverb, EAX reverb: slow_count), then
    // every other reverb instance of both types (general_offset / general_count).  list_count of a reverb type counts all its
    // instances.
    int list_offset[OALSFX_MAX_SLOTS][OALSFX_TYPE_COUNT] = {};
    int list_count[OALSFX_MAX_SLOTS][OALSFX_TYPE_COUNT] = {};
    int steady_offset[OALSFX_MAX_SLOTS] = {};     // start of the steady region: the proven instances, then the believed ones
    int fast_count[OALSFX_MAX_SLOTS] = {};        // proven steady (both reverb types)
    int kind_count[OALSFX_MAX_SLOTS][3] = {};     // ... of which, in list order: every tap two tiles away; a tap of one to two tiles; shorter taps or a
                                                  // modulated late line (the FP build each needs: plain, HY, ST)
    int rest_tiles[OALSFX_MAX_SLOTS] = {};        // ... whose output gains are at rest for blocks of that many tiles and longer (1 .. 4)
    int fast_first[OALSFX_MAX_SLOTS] = {};        // >= 0: the proven part of the list is the instance range fast_first, fast_first + 1, ...
    int slow_count[OALSFX_MAX_SLOTS] = {};        // believed steady (both reverb types)
    int general_offset[OALSFX_MAX_SLOTS] = {};
    int general_count[OALSFX_MAX_SLOTS] = {};
    std::map<size_t, RingPool> pools;
    std::vector<void*> chunks;
    struct RingChunk { char* base; size_t bytes; void* alloc; };
    std::vector<RingChunk> ring_chunks;            // the slab pools' chunks (for oalsfx_debug_move_rings)
    // placement search (place_ring_chunk): what it did for this batch, for benchmark records
    int placed_chunks = 0, placement_candidates = 0;
    double placement_best_us = 0.0, placement_worst_us = 0.0;

    // staging for host-pointer mixes
    float* d_io_src = nullptr;
    float* d_io_dst = nullptr;
    size_t io_capacity = 0;
    // ... and page-locked host buffers for callers whose instances each have buffers of their own (oalsfx_batch_mix_gather)
    float* h_io_src = nullptr;
    float* h_io_dst = nullptr;
    size_t h_io_capacity = 0;
    // ... and for the pipelined ones (oalsfx_batch_mix_async): kPipeDepth staging slots used in turn; the copy in, the kernels and the
    // copy out of successive calls run on three streams, ordered by events only
    static constexpr int kPipeDepth = 3;
    struct PipeSlot { float* d_src = nullptr; float* d_dst = nullptr; hipEvent_t copied_in = nullptr, mixed = nullptr, copied_out = nullptr; bool busy = false; };
    PipeSlot pipe[kPipeDepth];
    size_t pipe_capacity = 0;
    long long pipe_turn = 0;
    hipStream_t h2d_stream = nullptr, d2h_stream = nullptr;
    // How successive calls of oalsfx_batch_mix_async share the link: form 3: the copy in of call k + 1, the kernels of call k and the copy
    // out of call k - 1 on three streams, both directions busy at once; form 1: copy in, kernels and copy out of a call on the batch's one
    // stream, in order -- the synchronous call's sequence without its wait.  Some hosts' copy engines run the two directions far below the
    // link's rate when both are busy (round 3's driver box: 0.48 ms per step on three streams against 0.38 synchronous), on others three
    // streams are fastest (0.20 - 0.24 ms): decided per device by the first calls of the first batch that pipelines -- sixteen calls to
    // set everything up, sixteen on one stream, sixteen on three, the host's clock over the last twelve of either (the pipeline paces the caller: a call waits for
    // the one that used its staging slot three calls ago) -- and remembered; the single stream has to win by a tenth.
    // OALSFX_HOST_PIPELINE=1|3 fixes the form.  (Also built and measured, round 4: both copies on one stream in turn, the copy out of a
    // call queued behind the next call's copy in, the kernels beside them -- 0.41 ms where the synchronous call takes 0.37 and three
    // streams 0.24: a copy engine that changes direction every copy is slower than either.)
    int pipe_mode = 0;                      // 0: probing
    std::chrono::steady_clock::time_point pipe_probe_t0;
    double pipe_probe_us[2] = {0.0, 0.0};   // per call: what the probe saw on three streams / on one

    unsigned long long* d_timeline = nullptr;    // measurement only (OALSFX_DEBUG_TIMELINE=<file>)
    // Parameter uploads: the changed records are packed into pinned memory and put in place by one kernel (k_upload); four buffers
    // take turns so that the host never waits for the stream.
    struct Stage { char* host = nullptr; char* dev = nullptr; size_t capacity = 0; hipEvent_t done = nullptr; bool pending = false; };
    Stage stage[4];                               // (four: an event of this stack is seen complete only once the launch behind it has run,
                                                  // so the buffer of two uploads ago would still make the host wait for the GPU)
    int stage_turn = 0;
    hipEvent_t ev_uploaded = nullptr;             // parameter uploads of the batch's own stream -> a caller's launch stream
    hipEvent_t ev_mixed = nullptr;                // last launch on a caller's stream -> parameter uploads that overwrite what it reads
    hipStream_t last_launch_stream = nullptr;
    const char* last_steady_kernel = "";          // symbol of the last steady-state reverb launch

    hipStream_t stream = nullptr;
    // Chained launches: consecutive calls on the batch's own stream whose whole step is one steady-state reverb launch take turns on two
    // streams with nothing but a word per instance ordering them (KernelCtx::turn), so that the tail of one launch overlaps with the head
    // of the next (the ~6 us between dependent launches of one stream).  Off once the caller has asked for the stream handle: work
    // queued there by the caller expects the launches in stream order.
    hipStream_t chain_stream[kChainDepth] = {};   // [0] is `stream`; a run's launches take them in turn
    hipEvent_t ev_chain[kChainDepth] = {};
    int chain_pos = 0;                            // the stream of the last chained launch
    int chain_pos_last = -1, chain_pos_before = -1; // ... of the launch before this call's, and of the one before that (-1: none in this run)
    bool chain_used[kChainDepth] = {};            // streams the current run has launched on
    hipEvent_t ev_chain_start = nullptr;          // recorded in front of a run's first launch: the second (other stream) starts no earlier
    unsigned* d_turn = nullptr;                   // [n*slots], then the count of workgroups of chained launches that have started (k_chain_gate)
    uint32_t turn_counter = 0;                    // the number the last chained launch set
    uint32_t started_total = 0;                   // what that count comes to once the last chained launch has started as a whole
    int launched_groups = 0;                      // workgroups of the call's steady-state reverb launch (what its chained launch adds to that count)
    // Where the delay lines' write positions stand on the 128-byte line grid (32 frames of a ring line): a slot's position is the frames
    // mixed since its state was last started.  After a call that was not a multiple of 32 frames every ring store of a tile begins and
    // ends inside a line; the plain FP build then takes its line-aligned variant (reverb.hip, CR == 2).  A speed hint: the kernel goes by
    // the position it finds in the state.
    uint32_t frames_total = 0;                    // frames mixed by the batch so far (wraps)
    std::vector<uint32_t> started_at;             // [n*slots] frames_total when the slot's state was last started
    bool off_grid_known = false;                  // off_grid[] is up to date
    int off_grid[OALSFX_MAX_SLOTS] = {};          // per slot: reverbs whose write position is not a multiple of 32 frames
    bool poisoned = false;                        // a chained launch gave up waiting: some instance missed a buffer; every later call fails
    int chain_len = 0;                            // launches in the current run
    std::vector<std::pair<const char*, const char*>> chain_dsts; // ... and their output buffers
    bool uncached = false;                        // what launches hand on lives in uncached memory: calls can be chained launches
    bool chain_open = false;                      // the last call was a chained launch (its kernel may still run, on either stream)
    bool stream_handed_out = false;
    long long chained_calls = 0;
    // the kernel groups of one slot (ring-light effects, reverb, EAX reverb) touch disjoint instances: when more than one
    // is populated they run side by side on these streams, forked from and joined to the launch stream with events
    hipStream_t side_stream[kSideStreams] = {};
    hipEvent_t ev_fork = nullptr;
    hipEvent_t ev_join[kSideStreams] = {};
    const char* error = "";
    std::string error_store;

    bool timing = false;              // the launches of the current mix call carry events
    int timing_every = 0;             // 0: off; k: every k-th mix call is timed
    long long mix_calls = 0;
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> event_pool; // created when timing is switched on, so that the timed region creates none

    hipEvent_t take_event()
    {
        hipEvent_t e = nullptr;
        if (!event_pool.empty()) { e = event_pool.back(); event_pool.pop_back(); }
        else hipEventCreate(&e);
        return e;
    }
    bool fail(const char* msg) { error = msg; return false; }
    bool hip_ok(hipError_t e, const char* what)
    {
        if (e == hipSuccess) return true;
        error_store = std::string(what) + ": " + hipGetErrorString(e);
        error = error_store.c_str();
        return false;
    }
};

namespace {

constexpr const char* kErrRange = "Instance range is out of bounds.";
constexpr const char* kErrSlot = "Effect index is out of range.";   // reference ApiErrorMessages::effect_index_out_of_range
constexpr const char* kErrNoSrc = "No source samples.";              // reference ApiErrorMessages::no_src_samples
constexpr const char* kErrNoDst = "No destination samples.";         // reference ApiErrorMessages::no_dst_samples

bool range_ok(oalsfx_batch* b, int first, int count)
{
    if (first < 0 || count < 0 || first + count > b->n) return b->fail(kErrRange);
    return true;
}

void mark_touched(oalsfx_batch* b, int first, int count)
{
    for (int i = first; i < first + count; ++i)
        if (!b->touched[i]) {
            b->touched[i] = 1;
            b->touched_list.push_back(i);
        }
}

void mark_dirty(oalsfx_batch* b, int i)
{
    if (!b->inst_dirty[i]) {
        b->inst_dirty[i] = 1;
        b->dirty_list.push_back(i);
    }
}

void release_slab(oalsfx_batch* b, size_t idx)
{
    if (b->h_rings[idx]) {
        b->pools[b->ring_floats[idx]].free_dirty.push_back(b->h_rings[idx]);
        b->h_rings[idx] = nullptr;
        b->ring_floats[idx] = 0;
    }
}

int debug_flags(); // experiment / test switches, defined below

constexpr int kSettleFrames = OALSFX_RV_FADE_SAMPLES; // the cross-fade (128 frames) is over, and with it at least one call, whose end
                                                      // snaps the output gains to their targets (reference MixHelpers::mix)

// What the host remembers of a slot's parameters (a speed hint only: the kernels decide from the device state).
enum : uint8_t {
    kClassReverb = 1,  // reverb or EAX reverb
    kClassSteady = 2,  // parameters the steady-state kernel builds accept
    kClassClose = 4,   // shortest tap distance 64 .. kPlainMinTap - 1 samples, or an early tap / late-line offset under kPlainMinTapAhead (the HY builds)
    kClassShort = 8,   // shortest tap distance below 64 samples
    kClassModulated = 16,
};

uint8_t classify_slot(const oalsfx_slot_params& sp)
{
    if (sp.type != OALSFX_REVERB && sp.type != OALSFX_EAX_REVERB) return 0;
    const oalsfx_reverb_params& p = sp.u.reverb;
    uint8_t cls = kClassReverb | kClassSteady;
    const int sway = p.mod_depth != 0.0F ? 1 + static_cast<int>(std::abs(p.mod_depth)) : 0; // a modulated late line reads that much closer
    int lo = 1 << 30, lo_ahead = 1 << 30; // (lo_ahead: early taps and late-line offsets, kPlainMinTapAhead)
    for (int j = 0; j < 4; ++j) {
        // what the most general build of the steady-state kernel accepts: early / late taps and early-line offsets of any length,
        // all-pass offsets from four samples, late-line offsets from one tile
        if (p.early_tap[j] < 0 || p.early_ap_off[j] < 4 || p.early_line_off[j] < 0 || p.late_ap_off[j] < 4 ||
            p.late_line_off[j] < 64 + sway || p.late_tap[j] < p.late_feed_tap)
            cls &= static_cast<uint8_t>(~kClassSteady);
        lo = std::min({lo, p.early_tap[j], p.early_ap_off[j], p.early_line_off[j], p.late_tap[j] - p.late_feed_tap, p.late_ap_off[j], p.late_line_off[j]});
        lo_ahead = std::min({lo_ahead, p.early_tap[j], p.late_line_off[j]});
    }
    if (lo >= 64 && (lo < static_cast<int>(oalsfx_hip::kPlainMinTap) || lo_ahead < static_cast<int>(oalsfx_hip::kPlainMinTapAhead))) cls |= kClassClose;
    if (lo < 64) cls |= kClassShort;
    if (p.mod_depth != 0.0F) cls |= kClassModulated;
    return cls;
}

// Host-side belief about which reverb instances the steady-state kernel will fully process: they are listed first and go
// to that kernel, the others straight to the general kernel (a speed hint; the steady-state kernel decides on the device
// from the real state and falls back by itself).
bool reverb_settled(const oalsfx_batch* b, size_t idx)
{
    return (b->slot_class[idx] & kClassSteady) != 0 && (b->since_update[idx] >= kSettleFrames || b->xf_ok[idx]);
}

// Can the cross-fading build (reverb.hip, XF) take an instance from parameters `from` to parameters `to`?  Both tap sets must be ones
// the most general steady-state build accepts (classify_slot), the late taps counted from the *new* late feed position and the late
// line given the sway of whichever modulation is deeper.  A hint like the rest: the kernel checks against the device state and falls
// back by itself.
bool crossfade_followable(const oalsfx_reverb_params& from, const oalsfx_reverb_params& to)
{
    const int sway = (from.mod_depth != 0.0F || to.mod_depth != 0.0F) ? 1 + static_cast<int>(std::max(std::abs(from.mod_depth), std::abs(to.mod_depth))) : 0;
    for (const oalsfx_reverb_params* p : {&from, &to})
        for (int j = 0; j < 4; ++j)
            if (p->early_tap[j] < 0 || p->early_ap_off[j] < 4 || p->early_line_off[j] < 0 || p->late_tap[j] < to.late_feed_tap ||
                p->late_ap_off[j] < 4 || p->late_line_off[j] < 64 + sway)
                return false;
    return true;
}

// Is some reverb of the slot in a transition the cross-fading build follows?  (`settling` is short: the instances updated within the
// last 128 frames.)
bool slot_in_transition(const oalsfx_batch* b, int slot)
{
    for (size_t idx : b->settling)
        if (static_cast<int>(idx % b->slots) == slot && b->xf_ok[idx]) return true;
    return false;
}

void reclassify_slot(oalsfx_batch* b, size_t idx, int slot)
{
    const uint8_t before = b->slot_class[idx], after = classify_slot(b->h_params[idx]);
    b->n_close[slot] += ((after & kClassClose) != 0) - ((before & kClassClose) != 0);
    b->n_short[slot] += ((after & kClassShort) != 0) - ((before & kClassShort) != 0);
    if (after & kClassModulated) b->modulated[slot] = true;
    b->slot_class[idx] = after;
}

// After a mix: the slots that were updated recently move towards "steady"; the list order is stale once one arrives.
void advance_settling(oalsfx_batch* b, int frames)
{
    size_t keep = 0;
    for (size_t k = 0; k < b->settling.size(); ++k) {
        const size_t idx = b->settling[k];
        b->since_update[idx] = std::min(b->since_update[idx] + frames, kSettleFrames);
        if (b->since_update[idx] < kSettleFrames) { b->settling[keep++] = idx; continue; }
        b->in_settling[idx] = 0;
        b->xf_ok[idx] = 0;
        if (b->slot_class[idx] & kClassSteady) { b->lists_dirty = true; b->exact_wanted = true; }
    }
    b->settling.resize(keep);
}

// `chunks` zero-filled chunks of `count` delay-line slabs each, appended to `out`.  Where such a chunk lands in the card's memory is not
// the same everywhere: the reverb's ring traffic (48 streams per instance, a few hundred thousand in all) runs at one of three rates
// depending on the physical pages behind it, up to 15 % apart, about a third of the card each (scripts/vram_map.py,
// profiles/README.md) -- and the steady-state kernel follows the traffic-only probe to the microsecond
// (scripts/move_rings_probe.py).  So for chunks of 1 GiB and more the runtime does not take the first allocations it is given: it
// allocates candidates (all held, or the driver would hand the same pages out again), times the probe on each (k_ring_probe, which
// leaves the memory zero-filled), stops once it holds enough of the fastest kind and has seen a clearly slower one (or after
// 96 candidates -- four times the chunks wanted if that is more --, or half the free memory), keeps the fastest and frees the rest.  A few hundred milliseconds, once per
// batch.  OALSFX_DEBUG_FLAGS 0x4000000 switches the search off.
// What one launch hands to the next -- delay lines, effect state, hot records, send-filter histories, the turn words of chained launches
// -- lives, for a batch whose calls can be chained launches (chain_capable), in memory the L2s do not cache (hipDeviceMallocUncached): a
// wavefront's acknowledged stores are then in memory for every XCD to see, which is what lets consecutive calls overlap without a
// write-back of the L2 in between.  The kernels stream through this memory anyway: measured neutral on every workload
// (profiles/r03k_chained_launches/uncached_memory.txt: headline 42.9 / 43.1 / 42.4 us for default / fine-grained / uncached, the other
// BASELINE configurations within their noise).  Fine-grained memory is not enough: plain loads and stores of it are as stale across XCDs
// as those of ordinary memory (tests/test_gpu_chained.py fails on it).
// OALSFX_RING_MEMORY=default | finegrained | uncached forces one kind for every batch (comparisons; no chained launches unless uncached).
//
// Uncached memory is not given back to the runtime unless the process asks for it.  Round 3 found that uncached blocks freed with hipFree
// disturbed allocations made afterwards -- 512 bytes of zeros at the start of a page of a later batch's output, or a reverb instance off
// for a whole buffer -- and kept every block for the life of the process.  What round 4 established with the reproducer
// (scripts/uncached_free_hazard.py, profiles/r04b_uncached_free_hazard/):
//  - the buffer that reads wrong is the later batch's own staging buffer in *ordinary* memory (hipMalloc, the host-pointer entry point):
//    with the caller's buffers in device memory from another allocator (MODE=device) no run ever failed;
//  - its addresses do not lie in a range that was uncached before (hazard_trace.txt: 0 of 36 staging allocations), but the failing runs
//    are the ones whose staging buffers the runtime placed in the 8 MiB neighbourhood of the recycled uncached pages;
//  - nothing of this library is still in flight when a block is freed (every stream is waited for first, oalsfx_batch_destroy), and a
//    hipDeviceSynchronize directly in front of the hipFree changes nothing (hazard_modes.txt);
//  - with every uncached block a whole number of 2 MiB granules, and all of them freed, 4 of 4 runs were clean (granule.txt) -- but
//    with the small blocks in arenas that stay and only the delay lines freed, granules or not, it is back in 1 run of 4
//    (hazard_granules.txt): the size of the pieces is not the whole story.
// So: not a use after free of this library's, reproducible only through hipFree of hipDeviceMallocUncached memory followed by fresh
// ordinary allocations, and not understood further -- which is why the default stays "keep".  The blocks do come in 2 MiB granules now
// (small ones out of arenas of the pool's own: fewer runtime calls, and no 4 KiB uncached pieces among ordinary pages), what waits for
// reuse can be capped (OALSFX_UNCACHED_POOL_MAX_GIB: beyond it the largest waiting blocks are freed when a batch goes; unset: no limit),
// and oalsfx_trim_pools() frees everything that waits: for a process that needs the memory back and allocates nothing on the device
// afterwards that it cannot afford to check -- INTEGRATION.md says so.
class UncachedPool {
public:
    static constexpr size_t kGranule = size_t(2) << 20;
    static constexpr size_t kSmall = size_t(1) << 20; // blocks below this come out of arenas, 4 KiB apart
    hipError_t take(int device, size_t bytes, void** p)
    {
        if (bytes < kSmall) return take_small(device, bytes, p);
        bytes = (bytes + kGranule - 1) / kGranule * kGranule;
        {
            // the smallest block that is waiting, large enough and not more than twice as large (a process that goes through many batch
            // shapes then keeps a block per size class rather than per size)
            std::lock_guard<std::mutex> lock(mutex_);
            auto it = free_.lower_bound({device, bytes});
            if (it != free_.end() && it->first.first == device && it->first.second <= 2 * bytes) {
                *p = it->second;
                live_[*p] = it->first;
                waiting_ -= it->first.second;
                free_.erase(it);
                return hipSuccess;
            }
        }
        const hipError_t e = hipExtMallocWithFlags(p, bytes, hipDeviceMallocUncached);
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> lock(mutex_);
            live_[*p] = {device, bytes};
        }
        return e;
    }
    bool give_back(void* p) // false: not one of ours
    {
        std::vector<void*> to_free;
        {
            std::lock_guard<std::mutex> lock(mutex_);
            auto sm = small_.find(p);
            if (sm != small_.end()) {
                Arena& a = arenas_[sm->second];
                small_.erase(sm);
                if (--a.live == 0) a.used = 0; // (a bump allocator per arena: the batches that shared it are gone, it starts over)
                return true;
            }
            auto it = live_.find(p);
            if (it == live_.end()) return false;
            free_.insert({it->second, p});
            waiting_ += it->second.second;
            live_.erase(it);
            shrink_locked(cap(), to_free);
        }
        release(to_free);
        return true;
    }
    // Frees what waits for reuse down to `keep_bytes` (largest blocks first) and every arena nobody uses.  Returns the bytes freed.
    size_t trim(size_t keep_bytes)
    {
        std::vector<void*> to_free;
        size_t before = 0;
        {
            std::lock_guard<std::mutex> lock(mutex_);
            before = waiting_;
            shrink_locked(keep_bytes, to_free);
            before -= waiting_;
            if (keep_bytes == 0) {
                for (auto& a : arenas_)
                    if (a.base && a.live == 0) { to_free.push_back(a.base); a.base = nullptr; before += kGranule; }
            }
        }
        release(to_free);
        return before;
    }
    size_t bytes_waiting()
    {
        std::lock_guard<std::mutex> lock(mutex_);
        return waiting_;
    }

private:
    struct Arena { int device; char* base; size_t used; int live; };
    static size_t cap()
    {
        static const size_t c = [] {
            const char* e = std::getenv("OALSFX_UNCACHED_POOL_MAX_GIB");
            const double gib = e ? std::atof(e) : -1.0; // (unset: nothing is freed behind the caller's back)
            return gib < 0.0 ? ~size_t(0) : static_cast<size_t>(gib * 1073741824.0);
        }();
        return c;
    }
    hipError_t take_small(int device, size_t bytes, void** p)
    {
        bytes = (bytes + 4095) / 4096 * 4096; // (every block on pages of its own, as the runtime hands them out)
        std::lock_guard<std::mutex> lock(mutex_);
        for (size_t k = 0; k < arenas_.size(); ++k) {
            Arena& a = arenas_[k];
            if (a.base && a.device == device && a.used + bytes <= kGranule) {
                *p = a.base + a.used;
                a.used += bytes;
                a.live += 1;
                small_[*p] = k;
                return hipSuccess;
            }
        }
        void* base = nullptr;
        const hipError_t e = hipExtMallocWithFlags(&base, kGranule, hipDeviceMallocUncached);
        if (e != hipSuccess) return e;
        size_t k = 0;
        while (k < arenas_.size() && arenas_[k].base) ++k;
        if (k == arenas_.size()) arenas_.push_back({});
        arenas_[k] = {device, static_cast<char*>(base), bytes, 1};
        *p = base;
        small_[*p] = k;
        return hipSuccess;
    }
    void shrink_locked(size_t keep_bytes, std::vector<void*>& to_free)
    {
        while (waiting_ > keep_bytes && !free_.empty()) {
            auto largest = free_.begin();
            for (auto it = free_.begin(); it != free_.end(); ++it)
                if (it->first.second > largest->first.second) largest = it;
            waiting_ -= largest->first.second;
            to_free.push_back(largest->second);
            free_.erase(largest);
        }
    }
    static void release(const std::vector<void*>& blocks)
    {
        // (nothing of the library still uses a block that waits; the synchronisation is for whatever else the process has queued)
        if (blocks.empty()) return;
        (void)hipDeviceSynchronize();
        for (void* q : blocks) (void)hipFree(q);
    }
    std::mutex mutex_;
    std::multimap<std::pair<int, size_t>, void*> free_;
    std::map<void*, std::pair<int, size_t>> live_;
    std::vector<Arena> arenas_;
    std::map<void*, size_t> small_; // small block -> its arena
    size_t waiting_ = 0;
};

UncachedPool& uncached_pool()
{
    static UncachedPool* pool = new UncachedPool; // (never destroyed: batches may outlive static destruction order)
    return *pool;
}

// Does this runtime hand out uncached device memory at all?  Asked once per device; where it does not, batches keep everything in
// ordinary memory and their calls in stream order.
bool uncached_memory_available(int device)
{
    static std::mutex m;
    static std::map<int, bool> known;
    std::lock_guard<std::mutex> lock(m);
    auto it = known.find(device);
    if (it != known.end()) return it->second;
    // The hand-over between launches that run at the same time (reverb.hip) rests on how this part's caches behave -- an acknowledged
    // store to uncached memory is in memory for every XCD, a launch starts with empty vector L1s, HW_REG_XCC_ID names the XCD -- and has
    // been validated on gfx950 only: any other device keeps its calls in stream order (ADVICE, round 3).
    hipDeviceProp_t prop{};
    bool ok = hipGetDeviceProperties(&prop, device) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
    if (ok) {
        void* p = nullptr;
        ok = uncached_pool().take(device, 4096, &p) == hipSuccess;
        if (ok) uncached_pool().give_back(p);
        else (void)hipGetLastError();
    }
    known[device] = ok;
    return ok;
}

hipError_t handed_on_malloc(const oalsfx_batch* b, void** p, size_t bytes)
{
    const char* kind = std::getenv("OALSFX_RING_MEMORY");
    if (kind && std::strcmp(kind, "finegrained") == 0) return hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained);
    if (kind && std::strcmp(kind, "default") == 0) return hipMalloc(p, bytes);
    if (b->uncached || (kind && std::strcmp(kind, "uncached") == 0)) return uncached_pool().take(b->device, bytes, p);
    return hipMalloc(p, bytes);
}

void handed_on_free(void* p)
{
    if (p && !uncached_pool().give_back(p)) (void)hipFree(p);
}

bool place_ring_chunks(oalsfx_batch* b, int chunks, int count, size_t slab_floats, std::vector<float*>& out)
{
    if (chunks <= 0) return true;
    const size_t bytes = static_cast<size_t>(count) * slab_floats * sizeof(float);
    struct Candidate { void* p; double us; };
    std::vector<Candidate> cands;
    size_t free_b = 0, total_b = 0;
    // OALSFX_PLACEMENT=0 switches the search off (a process that shares the card with other allocators: the candidates are held until
    // the search ends); OALSFX_PLACEMENT_MAX_GIB caps what it may hold at once (default: half of the free memory)
    const char* env_on = std::getenv("OALSFX_PLACEMENT");
    const char* env_cap = std::getenv("OALSFX_PLACEMENT_MAX_GIB");
    const bool wanted = bytes >= (static_cast<size_t>(3) << 28) && slab_floats >= 65536 && !(debug_flags() & 0x4000000) && !(env_on && std::atoi(env_on) == 0);
    // one search at a time in a process: each budgets against the free memory it sees when it starts
    static std::mutex placement_mutex;
    std::unique_lock<std::mutex> placement_lock(placement_mutex, std::defer_lock);
    if (wanted) placement_lock.lock();
    size_t budget = 0;
    if (wanted && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        budget = free_b / 2;
        if (env_cap && std::atof(env_cap) > 0.0) budget = std::min(budget, static_cast<size_t>(std::atof(env_cap) * 1073741824.0));
    }
    const bool search = wanted && budget > bytes * (static_cast<size_t>(chunks) + 1);
    if (!search) {
        for (int k = 0; k < chunks; ++k) {
            void* p = nullptr;
            bool ok = b->hip_ok(handed_on_malloc(b, &p, bytes), "hipMalloc(rings)");
            if (ok) cands.push_back({p, 0.0});
            ok = ok && b->hip_ok(hipMemsetAsync(p, 0, bytes, b->stream), "hipMemsetAsync(rings)");
            if (!ok) {
                for (auto& c : cands) handed_on_free(c.p);
                return false;
            }
        }
    } else {
        const size_t max_tries = std::min<size_t>(std::max<size_t>(96, static_cast<size_t>(chunks) * 4), budget / bytes);
        const int waves_per_slab = std::max(1, 4096 / count); // a full load (4096 wavefronts) whatever the chunk's size
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (!b->hip_ok(hipEventCreate(&e0), "hipEventCreate") || !b->hip_ok(hipEventCreate(&e1), "hipEventCreate")) return false;
        bool ok = true;
        while (cands.size() < max_tries) {
            void* c = nullptr;
            if (handed_on_malloc(b, &c, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
            cands.push_back({c, 1e30});
            if (!(ok = b->hip_ok(hipMemsetAsync(c, 0, bytes, b->stream), "hipMemsetAsync(rings)"))) break;
            for (int r = 0; r < 2; ++r) oalsfx_hip::launch_ring_probe(static_cast<float*>(c), count, slab_floats, 256u * r, waves_per_slab, b->stream);
            hipEventRecord(e0, b->stream);
            for (int r = 0; r < 4; ++r) oalsfx_hip::launch_ring_probe(static_cast<float*>(c), count, slab_floats, 256u * (2 + r), waves_per_slab, b->stream);
            hipEventRecord(e1, b->stream);
            float ms = 0.0F;
            if (!(ok = b->hip_ok(hipEventSynchronize(e1), "hipEventSynchronize") && b->hip_ok(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime"))) break;
            cands.back().us = ms * 1e3 / 4;
            b->placement_candidates += 1;
            if (cands.size() < std::min<size_t>(max_tries, static_cast<size_t>(chunks) * 3 + 4)) continue; // a fair sample first
            std::vector<double> us;
            for (const auto& c2 : cands) us.push_back(c2.us);
            std::sort(us.begin(), us.end());
            // enough candidates of the fastest kind in hand, and a clearly slower kind seen next to them: done
            if (us[chunks - 1] <= us[0] * 1.04 && us[0] * 1.12 <= us.back()) break;
        }
        hipEventDestroy(e0); hipEventDestroy(e1);
        std::stable_sort(cands.begin(), cands.end(), [](const Candidate& x, const Candidate& y) { return x.us < y.us; });
        if (!ok || static_cast<int>(cands.size()) < chunks) {
            for (auto& c : cands) handed_on_free(c.p);
            return ok ? b->fail("hipMalloc(rings) failed") : false;
        }
        for (size_t k = chunks; k < cands.size(); ++k) handed_on_free(cands[k].p);
        b->placement_worst_us = std::max(b->placement_worst_us, cands.back().us);
        cands.resize(chunks);
        b->placement_best_us = cands.front().us;
        // (address order within the batch: consecutive instances then sit in consecutive slabs of consecutive chunks)
        std::sort(cands.begin(), cands.end(), [](const Candidate& x, const Candidate& y) { return x.p < y.p; });
    }
    for (auto& c : cands) {
        b->placed_chunks += 1;
        b->chunks.push_back(c.p);
        b->ring_chunks.push_back({static_cast<char*>(c.p), bytes, c.p});
        out.push_back(static_cast<float*>(c.p));
    }
    return true;
}

// Pinned staging buffer for one round of parameter uploads: four take turns, and a buffer is reused only once the launch that
// read it has run.
oalsfx_batch::Stage* acquire_stage(oalsfx_batch* b, size_t bytes)
{
    // the first buffer, oldest first, whose launch has run (a query: waiting for an event, however old, makes the host wait for
    // everything that is queued on this stack); only if all four are still being read does the host wait for the oldest
    int pick = -1;
    for (int k = 0; k < 4 && pick < 0; ++k) {
        oalsfx_batch::Stage& c = b->stage[(b->stage_turn + k) & 3];
        if (!c.pending || hipEventQuery(c.done) == hipSuccess) pick = (b->stage_turn + k) & 3;
    }
    if (pick < 0) {
        pick = b->stage_turn;
        const auto w0 = std::chrono::steady_clock::now();
        if (!b->hip_ok(hipEventSynchronize(b->stage[pick].done), "hipEventSynchronize")) return nullptr;
        b->host_stage_wait_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - w0).count();
    }
    oalsfx_batch::Stage& st = b->stage[pick];
    b->stage_turn = (pick + 1) & 3;
    st.pending = false;
    if (!st.done && !b->hip_ok(hipEventCreateWithFlags(&st.done, hipEventDisableTiming), "hipEventCreate")) return nullptr;
    if (bytes > st.capacity) {
        if (st.host) (void)hipHostFree(st.host);
        if (st.dev) (void)hipFree(st.dev);
        st.host = st.dev = nullptr;
        st.capacity = 0;
        const size_t cap = std::max<size_t>(bytes * 2, 1 << 20);
        if (!b->hip_ok(hipHostMalloc(reinterpret_cast<void**>(&st.host), cap), "hipHostMalloc(staging)")) return nullptr;
        if (!b->hip_ok(hipMalloc(reinterpret_cast<void**>(&st.dev), cap), "hipMalloc(staging)")) return nullptr;
        st.capacity = cap;
    }
    return &st;
}

// Folds all pending property changes into descriptors, device state and the launch plan: what the
// reference does lazily at the top of mix_data (update_context_sources, src/oalsfxpp.cpp:3397-3412)
// plus EffectSlot::set_effect's state re-creation (src/oalsfxpp.cpp:2688-2709).
// In two steps: prepare_params does the host's part and packs the staging buffer; launch_params puts it in place on a stream.
// (sync_params: both, on the batch's own stream, without waiting for it; `consumer` -- the stream the next launches go to -- is made to
// wait for the uploads with an event when it is a different stream.  mix_device looks at what was prepared before it decides where the
// call's launches go: chained launches.)
struct PendingUpload {
    bool any = false;                  // something had changed
    oalsfx_batch::Stage* st = nullptr; // nullptr: nothing to upload
    size_t bytes = 0;
    bool direct = true;                // read by the kernel straight from the page-locked buffer
    bool chainable = true;             // only records that can take their turn instance by instance (slot parameters, epochs) and lists
    oalsfx_hip::UploadJobs jobs{};
};

bool chain_join(oalsfx_batch* b);

bool prepare_params(oalsfx_batch* b, PendingUpload& pu)
{
    if (b->dirty_list.empty() && !b->lists_dirty) return true;
    pu.any = true;
    const size_t total = static_cast<size_t>(b->n) * b->slots;
    std::map<size_t, int> need; // size class -> slabs needed
    std::vector<size_t> restarted;
    std::vector<int> up_params, up_state, up_source, up_touched; // indices of the records to upload
    std::vector<uint32_t> up_epoch;                  // new epochs of the instances in up_touched
    bool any_type_change = false;
    if (!b->dirty_list.empty()) b->upload_gen += 1;

    // a caller's stream may still be running kernels that read what is about to be overwritten
    if (b->last_launch_stream && b->last_launch_stream != b->stream) {
        if (!b->hip_ok(hipStreamWaitEvent(b->stream, b->ev_mixed, 0), "hipStreamWaitEvent")) return false;
    }

    const auto hd0 = std::chrono::steady_clock::now();
    for (int i : b->dirty_list) {
        InstanceHost& h = b->inst[i];
        bool updated = false, sends_moved = false;
        for (int s = 0; s < b->slots; ++s) {
            if (!h.slot_changed[s]) continue;
            h.slot_changed[s] = false;
            updated = true;
            sends_moved |= h.slot_retyped[s]; // (the send to a slot depends on what the slot holds: nothing, or an effect)
            const size_t idx = static_cast<size_t>(i) * b->slots + s;
            oalsfx_slot_params& p = b->h_params[idx];
            // a settled reverb whose properties change without a change of type keeps its state and cross-fades: what the XF build follows
            const bool was_settled = (b->slot_class[idx] & kClassSteady) != 0 && b->since_update[idx] >= kSettleFrames && !h.slot_retyped[s];
            const oalsfx_reverb_params before = p.u.reverb;
            derive_slot(b->dev, h.active[s], p);
            p.update_seq = ++b->seq[idx];
            up_params.push_back(static_cast<int>(idx));
            reclassify_slot(b, idx, s);
            if (h.slot_retyped[s]) b->mod_ever[idx] = 0;
            if (b->slot_class[idx] & kClassModulated) b->mod_ever[idx] = 1;
            b->xf_ok[idx] = was_settled && (b->slot_class[idx] & kClassSteady) != 0 && b->channels <= 2 && !(debug_flags() & 0x40000000) &&
                            crossfade_followable(before, p.u.reverb);
            b->lists_dirty = true; // (it leaves the proven part of its list, or the steady part altogether)
            b->since_update[idx] = 0;
            if ((b->slot_class[idx] & kClassReverb) && !b->in_settling[idx]) {
                b->in_settling[idx] = 1;
                b->settling.push_back(idx);
            }
            if (h.slot_retyped[s]) {
                h.slot_retyped[s] = false;
                any_type_change = true;
                reset_slot_state(p.type, b->h_state_init[idx]);
                b->h_state_init[idx].seen_seq = p.update_seq - 1;
                up_state.push_back(static_cast<int>(idx));
                restarted.push_back(idx);
                b->started_at[idx] = b->frames_total;
                b->off_grid_known = false;
                release_slab(b, idx);
                const size_t floats = static_cast<size_t>(ring_floats_for(p.type, b->rate));
                if (floats) need[floats] += 1;
            }
        }
        if (h.source_changed) {
            h.source_changed = false;
            updated = true;
            sends_moved = true;
        }
        // The reference recomputes a source's sends whenever one of its slots' properties changed (update_context_sources,
        // src/oalsfxpp.cpp:3397-3412); their inputs -- send properties, which slots hold an effect -- did not move when only an effect's
        // own properties did, and the result is the same record: derived and uploaded only when they did.
        if (updated && sends_moved) {
            int types[OALSFX_MAX_SLOTS] = {};
            for (int s = 0; s < b->slots; ++s) types[s] = static_cast<int>(h.active[s].type_);
            oalsfx_source_params& sp = b->h_source[i];
            auto has_filter = [&](const oalsfx_source_params& x) {
                int any = x.direct.filter_type;
                for (int s = 0; s < b->slots; ++s)
                    if (x.aux[s].out_channels != 0) any |= x.aux[s].filter_type;
                return any != OALSFX_AF_NONE;
            };
            const bool before = has_filter(sp);
            derive_source(b->dev, b->slots, h.direct_props, h.aux_props, types, sp);
            b->n_filtered += static_cast<int>(has_filter(sp)) - static_cast<int>(before);
            b->inst_filtered[i] = has_filter(sp) ? 1 : 0;
            b->inside_version = ~0u;
            up_source.push_back(i);
        }
        if (updated) {
            up_touched.push_back(i);
            // whatever the device knew about this instance is out of date: its hot records (new epoch) and the proof that its reverbs
            // are steady (a send change alone leaves them steady, which the next read-back will confirm)
            up_epoch.push_back(++b->inst_epoch[i]);
            for (int s = 0; s < b->slots; ++s) {
                const size_t idx = static_cast<size_t>(i) * b->slots + s;
                if (b->proven[idx]) { b->proven[idx] = 0; b->lists_dirty = true; }
                b->updated_gen[idx] = b->upload_gen;
            }
            b->exact_wanted = true;
        }
        b->inst_dirty[i] = 0;
    }
    b->dirty_list.clear();
    b->host_derive_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - hd0).count();

    // slots that start over take and give back delay lines, which launches of a run of chained launches may still be at work on
    if (!restarted.empty() && !chain_join(b)) return false;

    // ring slabs: grow each size class once, zero fresh chunks in one memset
    for (auto& kv : need) {
        RingPool& pool = b->pools[kv.first];
        pool.slab_floats = kv.first;
        const int have = static_cast<int>(pool.free_clean.size() + pool.free_dirty.size());
        const int grow = kv.second - have;
        if (grow > 0) {
            // chunks of at most 1024 slabs (0.9 GiB of reverb delay lines), each placed on its own (place_ring_chunk); the slabs are
            // handed out in address order within a chunk, chunk after chunk, so that consecutive instances get consecutive slabs
            constexpr int kChunkSlabs = 1024; // 0.9 GiB of reverb delay lines
            std::vector<float*> bases;
            if (!place_ring_chunks(b, grow / kChunkSlabs, kChunkSlabs, kv.first, bases)) return false;
            const size_t full = bases.size();
            if (grow % kChunkSlabs && !place_ring_chunks(b, 1, grow % kChunkSlabs, kv.first, bases)) return false;
            std::vector<float*> fresh;
            for (size_t c = 0; c < bases.size(); ++c) {
                const int count = c < full ? kChunkSlabs : grow % kChunkSlabs;
                for (int k = 0; k < count; ++k) fresh.push_back(bases[c] + static_cast<size_t>(k) * kv.first);
            }
            for (size_t k = fresh.size(); k-- > 0;) pool.free_clean.push_back(fresh[k]);
        }
    }
    bool rings_changed = false;
    for (size_t idx : restarted) {
        const size_t floats = static_cast<size_t>(ring_floats_for(b->h_params[idx].type, b->rate));
        rings_changed = true; // the slab it held is gone from the table in any case
        if (!floats) continue;
        RingPool& pool = b->pools[floats];
        float* slab = nullptr;
        if (!pool.free_clean.empty()) {
            slab = pool.free_clean.back();
            pool.free_clean.pop_back();
        } else {
            slab = pool.free_dirty.back();
            pool.free_dirty.pop_back();
            if (!b->hip_ok(hipMemsetAsync(slab, 0, floats * sizeof(float), b->stream), "hipMemsetAsync(ring)")) return false;
        }
        b->h_rings[idx] = slab;
        b->ring_floats[idx] = floats;
    }

    // ---- the launch plan: one counting sort over the slots of every instance ----
    const bool rebuild_lists = any_type_change || b->lists_dirty;
    const auto hl0 = std::chrono::steady_clock::now();
    std::vector<int> lists;
    if (rebuild_lists) {
        lists.resize(total);
        // ring-light types, then the reverbs of both types: proven steady by kind (the FP build each needs: plain, close taps, short taps or
        // modulated), believed steady or in a transition the XF build follows, the others
        constexpr int kFast = OALSFX_REVERB, kSlow = OALSFX_REVERB + 3, kGeneral = OALSFX_REVERB + 4, kBuckets = OALSFX_REVERB + 5;
        std::vector<uint8_t> bucket_of(b->n);
        const bool force = (debug_flags() & 0x2000000) != 0; // test of the fault path only: every reverb counts as proven
        for (int s = 0; s < b->slots; ++s) {
            // one pass over the slot's instances: bucket, type counts, and the shortest block (in tiles) that leaves the gains of every
            // proven instance of the slot alone (calls whose last block is shorter do not take the FP builds)
            int count[kBuckets] = {};
            b->list_count[s][OALSFX_REVERB] = b->list_count[s][OALSFX_EAX_REVERB] = 0;
            b->rest_tiles[s] = 0;
            for (int i = 0; i < b->n; ++i) {
                const size_t idx = static_cast<size_t>(i) * b->slots + s;
                const int t = b->h_params[idx].type;
                int k = t;
                if (t >= OALSFX_REVERB) {
                    b->list_count[s][t] += 1;
                    const uint8_t cls = b->slot_class[idx];
                    if (!force && !reverb_settled(b, idx)) k = kGeneral;
                    else if (!force && !b->proven[idx]) k = kSlow;
                    else {
                        k = kFast + (((cls & kClassShort) || b->mod_ever[idx]) ? 2 : (cls & kClassClose) ? 1 : 0);
                        b->rest_tiles[s] = std::max<int>(b->rest_tiles[s], b->proven[idx]);
                    }
                }
                bucket_of[i] = static_cast<uint8_t>(k);
                count[k] += 1;
            }
            int start[kBuckets];
            int off = s * b->n;
            for (int k = 0; k < kBuckets; ++k) { start[k] = off; off += count[k]; }
            for (int t = 0; t < OALSFX_REVERB; ++t) { b->list_offset[s][t] = start[t]; b->list_count[s][t] = count[t]; }
            for (int t = 0; t < 2; ++t) b->list_offset[s][OALSFX_REVERB + t] = start[kFast]; // the reverb region as a whole (the two types interleave)
            b->steady_offset[s] = start[kFast];
            for (int k = 0; k < 3; ++k) b->kind_count[s][k] = count[kFast + k];
            b->fast_count[s] = count[kFast] + count[kFast + 1] + count[kFast + 2];
            b->slow_count[s] = count[kSlow];
            b->general_offset[s] = start[kGeneral];
            b->general_count[s] = count[kGeneral];
            int fill[kBuckets];
            for (int k = 0; k < kBuckets; ++k) fill[k] = start[k];
            for (int i = 0; i < b->n; ++i) lists[fill[bucket_of[i]]++] = i;
            // a proven part that is simply a range of instances needs no list on the device
            b->fast_first[s] = b->fast_count[s] > 0 ? lists[start[kFast]] : -1;
            for (int k = 1; k < b->fast_count[s]; ++k)
                if (lists[start[kFast] + k] != b->fast_first[s] + k) { b->fast_first[s] = -1; break; }
        }
        b->lists_dirty = false;
        b->h_lists = lists;
        b->lists_version += 1;
    }

    b->host_lists_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - hl0).count();
    // ---- one packed upload: [indices | records] per array, the ring table, the lists ----
    auto padded = [](size_t v) { return (v + 15) & ~static_cast<size_t>(15); };
    const size_t n_p = up_params.size(), n_s = up_state.size(), n_src = up_source.size(), n_t = up_touched.size();
    size_t off = 0;
    const size_t o_pi = off; off += padded(n_p * sizeof(int));
    const size_t o_pr = off; off += padded(n_p * sizeof(oalsfx_slot_params));
    const size_t o_si = off; off += padded(n_s * sizeof(int));
    const size_t o_sr = off; off += padded(n_s * sizeof(oalsfx_hip::SlotStateLines)); // (records in whole lines, as they lie in device memory)
    const size_t o_ci = off; off += padded(n_src * sizeof(int));
    const size_t o_cr = off; off += padded(n_src * sizeof(oalsfx_source_params));
    const size_t o_ti = off; off += padded(n_t * sizeof(int));
    const size_t o_ep = off; off += padded(n_t * sizeof(uint32_t));
    const size_t o_rt = off; off += rings_changed ? padded(total * sizeof(float*)) : 0;
    const size_t o_li = off; off += rebuild_lists ? padded(total * sizeof(int)) : 0;
    if (off > 0) {
        oalsfx_batch::Stage* st = acquire_stage(b, off);
        if (!st) return false;
        if (n_p) std::memcpy(st->host + o_pi, up_params.data(), n_p * sizeof(int));
        for (size_t k = 0; k < n_p; ++k) std::memcpy(st->host + o_pr + k * sizeof(oalsfx_slot_params), &b->h_params[up_params[k]], sizeof(oalsfx_slot_params));
        if (n_s) std::memcpy(st->host + o_si, up_state.data(), n_s * sizeof(int));
        for (size_t k = 0; k < n_s; ++k) std::memcpy(st->host + o_sr + k * sizeof(oalsfx_hip::SlotStateLines), &b->h_state_init[up_state[k]], sizeof(oalsfx_slot_state));
        if (n_src) std::memcpy(st->host + o_ci, up_source.data(), n_src * sizeof(int));
        for (size_t k = 0; k < n_src; ++k) std::memcpy(st->host + o_cr + k * sizeof(oalsfx_source_params), &b->h_source[up_source[k]], sizeof(oalsfx_source_params));
        if (n_t) std::memcpy(st->host + o_ti, up_touched.data(), n_t * sizeof(int));
        if (n_t) std::memcpy(st->host + o_ep, up_epoch.data(), n_t * sizeof(uint32_t));
        if (rings_changed) std::memcpy(st->host + o_rt, b->h_rings.data(), total * sizeof(float*));
        if (rebuild_lists) std::memcpy(st->host + o_li, lists.data(), total * sizeof(int));
        // Small uploads (parameter changes while streaming) are read by the kernels straight from the page-locked staging buffer: a
        // copy-engine transfer between two kernels of one stream is ordered through the host on this stack and started ~90 us after
        // the kernel before it had ended (rocprofv3 trace of scripts/update_storm_bench.py), with the host waiting for it a step later.
        // Bulk uploads (creation of a batch) keep the transfer.
        const bool direct = off <= (static_cast<size_t>(1) << 20) && !(debug_flags() & 0x10000000);
        const char* from = direct ? st->host : st->dev;
        // one launch puts everything in place
        auto words = [&](size_t o) { return reinterpret_cast<const unsigned*>(from + o); };
        auto ints = [&](size_t o) { return reinterpret_cast<const int*>(from + o); };
        if (rebuild_lists) b->d_lists = b->d_lists_buf[b->lists_turn = (b->lists_turn + 1) % (kChainDepth + 1)]; // (launches in flight keep the list they were given)
        oalsfx_hip::UploadJobs& jobs = pu.jobs;
        jobs = oalsfx_hip::UploadJobs{};
        // (slot parameters and epochs take their turn instance by instance when the upload runs beside the launch before it)
        jobs.scatter[0] = {reinterpret_cast<unsigned*>(b->d_params), words(o_pr), ints(o_pi), static_cast<int>(sizeof(oalsfx_slot_params) / 4), static_cast<int>(n_p), 1};
        jobs.scatter[1] = {reinterpret_cast<unsigned*>(b->d_state), words(o_sr), ints(o_si), static_cast<int>(sizeof(oalsfx_hip::SlotStateLines) / 4), static_cast<int>(n_s), 0};
        jobs.scatter[2] = {reinterpret_cast<unsigned*>(b->d_source), words(o_cr), ints(o_ci), static_cast<int>(sizeof(oalsfx_source_params) / 4), static_cast<int>(n_src), 0};
        jobs.scatter[3] = {reinterpret_cast<unsigned*>(b->d_inst_epoch), words(o_ep), ints(o_ti), 1, static_cast<int>(n_t), 1};
        jobs.copy[0] = {reinterpret_cast<unsigned*>(b->d_rings), words(o_rt), rings_changed ? total * (sizeof(float*) / 4) : 0, 0};
        jobs.copy[1] = {reinterpret_cast<unsigned*>(b->d_lists), words(o_li), rebuild_lists ? total : 0, 0};
        pu.st = st;
        pu.bytes = off;
        pu.direct = direct;
        pu.chainable = direct && n_s == 0 && n_src == 0 && !rings_changed && need.empty();
    }
    return true;
}

// `turn` / `turn_wait`: the upload runs beside a chained launch that may still be at work (see UploadJobs).
bool launch_params(oalsfx_batch* b, PendingUpload& pu, hipStream_t upload_stream, hipStream_t consumer, const unsigned* turn = nullptr, unsigned turn_wait = 0)
{
    if (pu.st) {
        oalsfx_batch::Stage* st = pu.st;
        pu.st = nullptr;
        if (!pu.direct && !b->hip_ok(hipMemcpyAsync(st->dev, st->host, pu.bytes, hipMemcpyHostToDevice, upload_stream), "hipMemcpyAsync(parameters)")) return false;
        pu.jobs.turn = turn;
        pu.jobs.turn_wait = turn_wait;
        pu.jobs.fault = b->d_fault;
        oalsfx_hip::launch_upload(pu.jobs, upload_stream);
        // the staging buffer is free again once everything that reads it has run
        if (!b->hip_ok(hipEventRecord(st->done, upload_stream), "hipEventRecord")) return false;
        st->pending = true;
        if (!b->hip_ok(hipGetLastError(), "parameter upload")) return false;
    }
    // a launch stream other than the one the uploads ran on must see them
    if (pu.any && consumer && consumer != upload_stream) {
        if (!b->hip_ok(hipEventRecord(b->ev_uploaded, upload_stream), "hipEventRecord")) return false;
        if (!b->hip_ok(hipStreamWaitEvent(consumer, b->ev_uploaded, 0), "hipStreamWaitEvent")) return false;
    }
    return true;
}

bool sync_params(oalsfx_batch* b, hipStream_t consumer)
{
    PendingUpload pu;
    return prepare_params(b, pu) && launch_params(b, pu, b->stream, consumer);
}

bool ensure_mixbuf(oalsfx_batch* b)
{
    if (b->d_mixbuf || b->slots == 1) return true;
    const size_t bytes = static_cast<size_t>(b->n) * b->channels * OALSFX_MAX_CHUNK * sizeof(float);
    // (handed from the reverb-free slots' launch to the reverbs' in a chained step: with the rest of what launches hand on)
    return b->hip_ok(handed_on_malloc(b, reinterpret_cast<void**>(&b->d_mixbuf), bytes), "hipMalloc(mixbuf)");
}

// Timing experiments and test switches (OALSFX_DEBUG_FLAGS, or oalsfx_debug_set_flags for A/B runs inside one process): 1 / 2 / 4 the
// hand-over of chained launches (reverb.hip: every wavefront pays for the acquire behind its wait / none does / instance lines read
// before the turn has come: tests/test_gpu_chained.py), 8 every reverb through the
// general kernel, 32 / 64 tap distances rounded to 128 / 256 bytes in the steady-state kernel (results wrong on purpose,
// scripts/ablate_align.sh), 0x20000 no side streams, 0x80000 no mixed grid (ring-light effects and steady reverbs of a slot as two
// launches), 0x100000 no cooperative workgroups for the ring-light effects, 0x200000 no proven-steady builds (proven instances go
// through the believing builds), 0x400000 ring-light workgroups longest type first instead of in list (type) order, 0x800000 oalsfx_batch_mix_async copies page-locked
// buffers with kernels instead of the runtime's copy engines, 0x2000000 every reverb listed as proven steady whatever the device said
// (exercises the fault counter of the FP builds: tests only), 0x4000000 no placement search for the delay-line chunks, 0x8000000 no
// fused runs of reverb-free slots (one launch per slot; config 3: 118.4 against 107.5 us per step), 0x10000000 small parameter
// uploads through the copy engine like bulk ones (update storm, 4 changes per buffer: 216 against 170 us per step), 0x20000000 the
// caller's stream takes a slot's first part instead of its general kernel, 0x40000000 no cross-fading build: reverbs whose properties
// change go to the general kernel for 128 frames, as before round 3, 0x1000000 the believed kind of k_reverb_steady_kinds without the
// general path inside (experiment: what the fallback's scratch frame costs the grid), 0x200 no send filters inside the steady-state
// builds (the pre-pass kernel for every filtered instance, as before round 3), 0x400 no chained launches: consecutive calls in plain
// stream order (bench.py --no-chain), 0x8000 chained launches for short calls of small batches too (tests of the hand-over with few
// workgroups; measured slower: chain_eligible), 0x2000 no proven ragged builds (calls that end in a partial tile on the believing build, in stream
// order: as before round 4), 0x4000 no line-aligned store build for write positions off the line grid (reverb.hip, CR == 2: as
// before round 4), 0x800 the gate of chained launches in front of a run's second launch only (a negative control
// of tests/test_gpu_chained.py: it must fail), 0x40000 none of the shapes that chain since late round 4 (steps of two launches, the mixed grid, more than two channels: stream
// order as before), 0x10 a chained step's ring-light launch in its own list order instead of the reverbs' grid's (experiment),
// 0x80 batches without any reverb chain their calls too (tests of the ring-light kernel's hand-over; measured slower: chain_eligible),
// 0x1000 a chained step's two kernels with the workgroup sizes they declare (experiment: the places one kernel's workgroups give up
// do not fit the other's).  Environment beside the flags: OALSFX_RING_MEMORY=default|finegrained|uncached (where
// delay lines, state and hot records live), OALSFX_HOST_PROFILE (what the host spends in prepare_params, printed by synchronize)
std::atomic<int> g_debug_flags{-1}; // process-wide, read by every batch on whatever host thread drives it
int debug_flags()
{
    int f = g_debug_flags.load(std::memory_order_relaxed);
    if (f < 0) {
        const char* env = std::getenv("OALSFX_DEBUG_FLAGS");
        f = env ? static_cast<int>(std::strtol(env, nullptr, 0)) & 0x7FFFFFFF : 0; // decimal or 0x...
        g_debug_flags.store(f, std::memory_order_relaxed);
    }
    return f;
}

// Brackets one kernel launch with events recorded on its stream.  (Events attached to the launch itself through
// hipExtLaunchKernel were tried: they read 5-6 us longer than rocprofv3's kernel trace of the same run, these 1-2 us.)
struct ScopedTiming {
    oalsfx_batch* b; hipStream_t stream; TimedLaunch tl{};
    ScopedTiming(oalsfx_batch* b_, int type, hipStream_t s) : b(b_), stream(s)
    {
        if (!b->timing) return;
        tl.start = b->take_event();
        tl.stop = b->take_event();
        tl.type = type;
        hipEventRecord(tl.start, stream);
    }
    ~ScopedTiming()
    {
        if (!b->timing) return;
        hipEventRecord(tl.stop, stream);
        b->timed.push_back(tl);
    }
};

constexpr size_t kTimelineBytes = (64 * 4 + 64) * 96 * sizeof(unsigned long long); // 64 sampled workgroups x 4 waves x 96 stamps (steady-state kernel) + 64 sampled instances x 96 (general path)
constexpr int kTimedGeneralOffset = 16; // TimedLaunch::type of a reverb type's general-kernel launches
constexpr int kTimedWaveEffects = -1; // TimedLaunch::type of the merged launch for the ring-light effect types
constexpr int kTimedMixed = -2;       // ... of the grid that serves ring-light effects and steady reverbs of a slot together

// Can the steady-state kernel be used for this chunk at all?
bool steady_kernel_usable(const KernelCtx& ctx) { return ctx.frames >= 1 && !(debug_flags() & 8); } // any call size: a short call is one partial tile

// Does some reverb of the slot write its delay lines off the 128-byte line grid?  (Recounted after a call that was not a multiple of 32
// frames and after a slot restarted: every position moves with every call, by the same amount.)
bool slot_off_grid(oalsfx_batch* b, int slot)
{
    if (!b->off_grid_known) {
        for (int s = 0; s < b->slots; ++s) b->off_grid[s] = 0;
        const size_t total = static_cast<size_t>(b->n) * b->slots;
        for (size_t idx = 0; idx < total; ++idx)
            if ((b->slot_class[idx] & kClassReverb) && ((b->frames_total - b->started_at[idx]) & 31u) != 0u) b->off_grid[idx % b->slots] += 1;
        b->off_grid_known = true;
    }
    return b->off_grid[slot] > 0 && !(debug_flags() & 0x4000);
}

// One steady-state launch for steady instances of both reverb types (adjacent in the list; the kernel reads the type per
// instance): the proven ones (`proven`: the builds without steady-state test and general path), the believed ones, or both
// together through the believing builds.
void launch_reverb_steady_part(oalsfx_batch* b, const KernelCtx& ctx, int slot, int flags, int offset, int count, bool proven, hipStream_t stream)
{
    const int* list = b->d_lists + offset;
    // Mono / stereo: one launch, an instance that turns out not to be steady falls back inside it.  More than two channels:
    // the steady-state kernel notes per instance whether it took it, and the general kernel right behind it does the others.
    // (A chunk that is not a whole number of 64-frame tiles ends in a partial tile inside the steady-state kernel.)
    const bool hand_over = ctx.channels > 2;
    KernelCtx c = ctx;
    c.progress = hand_over ? b->d_progress : nullptr;
    c.list_first = proven ? b->fast_first[slot] : -1;
    // the proven-steady instances lead the steady region of the list: in a call of whole tiles (what "proven" vouches for) the
    // general kernel need not look at them
    // (whose last, shortest block is long enough for the gains of all of them to be at rest)
    const int n = ctx.frames;
    const bool vouched = (n & 63) == 0 && (n - ((n - 1) / OALSFX_RV_MAX_UPDATE) * OALSFX_RV_MAX_UPDATE) / 64 >= b->rest_tiles[slot];
    const int lead = hand_over && vouched && !(debug_flags() & 0x200000) ? std::max(0, std::min(count, b->steady_offset[slot] + b->fast_count[slot] - offset)) : 0;
    c.no_follow_up = lead;
    {
        ScopedTiming timing(b, OALSFX_EAX_REVERB, stream);
        int groups = 0;
        const char* name = oalsfx_hip::launch_reverb_steady(c, slot, list, count, flags | ((debug_flags() & 0xFF) << 8), b->n_close[slot] > 0,
                                                            b->modulated[slot], b->n_short[slot] > 0, proven, !proven && slot_in_transition(b, slot), stream, &groups,
                                                            proven && slot_off_grid(b, slot));
        if (name) b->last_steady_kernel = name;
        b->launched_groups += groups;
    }
    if (hand_over && count > lead) {
        ScopedTiming timing(b, OALSFX_REVERB + kTimedGeneralOffset, stream);
        oalsfx_hip::launch_reverb_general(c, slot, list + lead, count - lead, flags, stream);
    }
}

// Mono / stereo, whole tiles: every steady reverb of the slot in one launch, each kind on the build it needs (the proven ones by their
// taps: plain, HY or ST FP build; the believed ones and those in a transition the XF build follows on that build).  `proven_usable`:
// the call's last block leaves the gains of the proven instances at rest; otherwise they go with the believed ones for this call (the
// XF build ramps gains).
void steady_kind_counts(const oalsfx_batch* b, int slot, bool proven_usable, int counts[4])
{
    counts[0] = counts[1] = counts[2] = 0;
    counts[3] = b->slow_count[slot];
    if (proven_usable) for (int k = 0; k < 3; ++k) counts[k] = b->kind_count[slot][k];
    else counts[3] += b->fast_count[slot];
    // Every kind starts a new workgroup, and 4096 instances are exactly the 1024 workgroups the chip holds at once: one workgroup more
    // runs behind all the others and takes the launch from 52 to 70 us (measured).  So a kind's last incomplete workgroup goes to the
    // next populated kind instead, whose build is the more general one (plain < HY < ST; the XF build takes any steady instance): the
    // boundary in the list moves down by up to three entries.
    for (int k = 0; k < 3; ++k) {
        int next = k + 1;
        while (next < 4 && counts[next] == 0) ++next;
        if (next == 4 || counts[k] == 0) continue;
        const int carry = counts[k] & 3;
        counts[k] -= carry;
        counts[next] += carry;
    }
}

// Single-slot batches whose steady reverbs go by kind: the first two kinds' builds filter their own sends (SF, reverb.hip), the pre-pass
// covers the rest of the list.  Returns how many filtered instances those two kinds hold (0: no SF builds this call).
int filters_inside_count(oalsfx_batch* b, int slot, const int counts[4])
{
    if (b->slots != 1 || b->channels > 2 || b->n_filtered == 0 || counts[0] + counts[1] == 0 || (debug_flags() & 0x200)) return 0;
    const int end = counts[0] + counts[1];
    if (b->inside_version != b->lists_version || b->inside_end != end) {
        int inside = 0;
        const int* list = b->h_lists.data() + b->steady_offset[slot];
        for (int k = 0; k < end; ++k) inside += b->inst_filtered[list[k]];
        b->inside_version = b->lists_version;
        b->inside_end = end;
        b->inside_filtered = inside;
    }
    return b->inside_filtered;
}

void launch_reverb_kinds_part(oalsfx_batch* b, const KernelCtx& ctx, int slot, int flags, bool proven_usable, bool filters_inside, hipStream_t stream)
{
    int counts[4];
    steady_kind_counts(b, slot, proven_usable, counts);
    KernelCtx c = ctx;
    c.progress = nullptr;
    // one proven kind alone: its list may be a plain range of instances
    const bool one_kind = counts[3] == 0 && (counts[0] > 0) + (counts[1] > 0) + (counts[2] > 0) == 1;
    c.list_first = one_kind ? b->fast_first[slot] : -1;
    ScopedTiming timing(b, OALSFX_EAX_REVERB, stream);
    int groups = 0;
    const char* name = oalsfx_hip::launch_reverb_steady_kinds(c, slot, b->d_lists + b->steady_offset[slot], counts,
                                                              flags | ((debug_flags() & 0xFF) << 8) | ((debug_flags() & 0x100) ? oalsfx_hip::kNoCuMajor : 0),
                                                              (debug_flags() & 0x1000000) != 0, filters_inside, stream, &groups, slot_off_grid(b, slot));
    if (name) b->last_steady_kernel = name;
    b->launched_groups += groups;
}

// The general kernel takes the instances of both reverb types that are not believed steady, or every reverb instance of
// the slot when the steady-state kernel cannot be used for this chunk (the regions are adjacent in the list).
void launch_reverb_general_part(oalsfx_batch* b, bool everything, const KernelCtx& ctx, int slot, int flags, hipStream_t stream)
{
    const int offset = everything ? b->steady_offset[slot] : b->general_offset[slot];
    const int count = b->general_count[slot] + (everything ? b->fast_count[slot] + b->slow_count[slot] : 0);
    ScopedTiming timing(b, OALSFX_REVERB + kTimedGeneralOffset, stream);
    oalsfx_hip::launch_reverb_general(ctx, slot, b->d_lists + offset, count, flags, stream);
}

// Effect types whose filter recurrences gain from cooperative workgroups (wave_effects_body.hpp, chain_phase).
bool cooperative_type(int type)
{
    // (the compressor's follower goes through chain_phase too, but a wavefront's own is as fast: it is short, and what a
    // workgroup saves in instructions it loses at the two barriers)
    return type == OALSFX_DISTORTION || type == OALSFX_ECHO || type == OALSFX_EQUALIZER || type == OALSFX_RING_MODULATOR;
}

// The ring-light types of a slot from `first_type` on as segments of one grid: per type the whole workgroups (cooperative
// where the type gains from it), then the up to three instances left over.  Returns the number of instances covered.
int wave_segments(const oalsfx_batch* b, int slot, int first_type, oalsfx_hip::WaveSegments& seg)
{
    seg = oalsfx_hip::WaveSegments{};
    const bool coop_allowed = !(debug_flags() & 0x100000);
    const bool longest_first = (debug_flags() & 0x400000) != 0; // experiment: measured 64.3 against 62.8 us per step on config 4 in list order
    // how long a workgroup of the type runs, relative (4096 instances of one type, 256-frame buffers, profiles/: microseconds)
    static const int kCost[OALSFX_REVERB] = {8, 15, 26, 10, 10, 40, 21, 25, 15, 18}; // null, chorus, compressor, dedicated x 2, distortion, echo, equalizer, flanger, ring modulator
    struct Part { int count, offset, cost; bool coop; };
    Part parts[oalsfx_hip::WaveSegments::kMax];
    int n = 0, total = 0;
    auto add = [&](int count, bool coop, int cost) {
        if (count <= 0) return;
        parts[n++] = Part{count, total, cost, coop};
        total += count;
    };
    for (int t = first_type; t < OALSFX_REVERB; ++t) {
        const int count = b->list_count[slot][t];
        if (coop_allowed && cooperative_type(t) && count >= 4) {
            add(count & ~3, true, kCost[t]);
            add(count & 3, false, kCost[t]);
        } else {
            add(count, false, kCost[t]);
        }
    }
    if (longest_first) std::stable_sort(parts, parts + n, [](const Part& x, const Part& y) { return x.cost > y.cost; });
    for (int k = 0; k < n; ++k) {
        seg.count[k] = parts[k].count;
        seg.offset[k] = parts[k].offset;
        if (parts[k].coop) seg.coop_mask |= 1u << k;
    }
    seg.n = n;
    return total;
}

// All ring-light effect types of a slot in one grid: their instance lists are adjacent in d_lists (types in
// ascending order, the two reverb types last).
void launch_wave_group(oalsfx_batch* b, const KernelCtx& ctx, int slot, int flags, hipStream_t stream)
{
    const bool null_has_duty = (flags & (oalsfx_hip::kFirst | oalsfx_hip::kLast)) != 0;
    const int first_type = null_has_duty ? OALSFX_NULL : OALSFX_NULL + 1;
    oalsfx_hip::WaveSegments seg;
    const int count = wave_segments(b, slot, first_type, seg);
    if (count == 0) return;
    ScopedTiming timing(b, kTimedWaveEffects, stream);
    if (ctx.turn != nullptr) { flags |= (debug_flags() & 7) << 8; b->launched_groups += seg.blocks(); } // (a chained launch: test switches, the gate's count)
    oalsfx_hip::launch_wave_effects(ctx, slot, 1, b->d_lists + b->list_offset[slot][first_type], count, &seg, flags, stream);
}

// Do the reverb groups of a slot's mixed grid take the proven build for a call of n frames?  Every reverb of the slot proven, whole
// tiles, and a last block that leaves their gains at rest.
bool mixed_grid_proven(const oalsfx_batch* b, int slot, int n)
{
    return b->slow_count[slot] == 0 && b->fast_count[slot] > 0 && (n & 63) == 0 &&
           (n - ((n - 1) / OALSFX_RV_MAX_UPDATE) * OALSFX_RV_MAX_UPDATE) / 64 >= b->rest_tiles[slot] && !(debug_flags() & 0x200000);
}

// Ring-light effects and believed-steady reverbs of one slot in one grid (k_slot_mixed).
void launch_mixed_part(oalsfx_batch* b, const KernelCtx& ctx, int slot, int flags, hipStream_t stream)
{
    const bool null_has_duty = (flags & (oalsfx_hip::kFirst | oalsfx_hip::kLast)) != 0;
    const int first_type = null_has_duty ? OALSFX_NULL : OALSFX_NULL + 1;
    oalsfx_hip::WaveSegments seg;
    const int light = wave_segments(b, slot, first_type, seg);
    const int steady = b->fast_count[slot] + b->slow_count[slot];
    KernelCtx c = ctx;
    c.progress = nullptr; // an instance that turns out not to be steady falls back inside the grid
    // all of them proven, and a call whose last block leaves their gains at rest: the reverb groups take the FP build
    const bool proven = mixed_grid_proven(b, slot, ctx.frames);
    c.list_first = proven ? b->fast_first[slot] : -1;
    ScopedTiming timing(b, kTimedMixed, stream);
    if (ctx.turn != nullptr) flags |= (debug_flags() & 0xFF) << 8; // (a chained launch: the hand-over's test switches)
    int groups = 0;
    oalsfx_hip::launch_slot_mixed(c, slot, b->d_lists + b->steady_offset[slot], steady, b->d_lists + b->list_offset[slot][first_type], light,
                                  seg, flags, proven, stream, &groups);
    b->launched_groups += groups;
}

// Number of consecutive slots from `slot` on that hold no reverb at all: such a run is one fused launch over every instance.
int reverb_free_run(const oalsfx_batch* b, int slot)
{
    int n = 0;
    while (slot + n < b->slots && b->list_count[slot + n][OALSFX_REVERB] + b->list_count[slot + n][OALSFX_EAX_REVERB] == 0) ++n;
    return n;
}

// A finished read-back of the "settled and at rest" flags turns believed-steady reverbs into proven ones.  Never waits: an event that
// has not completed yet is looked at again by the next call.
void poll_exact(oalsfx_batch* b)
{
    if (!b->exact_pending || hipEventQuery(b->ev_exact) != hipSuccess) return;
    b->exact_pending = false;
    const size_t total = static_cast<size_t>(b->n) * b->slots;
    for (size_t idx = 0; idx < total; ++idx) {
        // the flag describes the slot as of the read-back's call: usable when nothing was uploaded for the instance after that
        if (b->proven[idx] || !b->h_exact[idx] || b->updated_gen[idx] > b->exact_gen || !reverb_settled(b, idx)) continue;
        b->proven[idx] = static_cast<uint8_t>(std::min(4u, (b->h_exact[idx] + 63u) / 64u)); // the rest level in tiles: 1 .. 4
        b->lists_dirty = true;
    }
}

// Reported by the calls that wait for the stream: a proven-steady launch found an instance that was not steady (it left it
// unprocessed).  The host's bookkeeping makes that impossible; if it happens anyway it must not pass silently.
bool check_fault(oalsfx_batch* b)
{
    if (!b->h_fault || (b->h_fault[0] == 0 && b->h_fault[1] == 0)) return true;
    const unsigned f = b->h_fault[0], gates = b->h_fault[1] / oalsfx_hip::kFaultGate;
    if (f == 0) {
        // Gates gave up waiting and nothing else went wrong: every launch behind them found its instances' turns (else the word above would
        // say so), the results are whole.  Something holds launches of one queue back until kernels of another have finished -- a tool
        // that runs kernels one at a time out of queue order -- and every gate costs its full wait (1.3 s).  The batch goes on in plain
        // stream order from here, and says so once.
        b->h_fault[1] = 0;
        if (!b->chain_given_up) {
            b->chain_given_up = true;
            std::fprintf(stderr, "oalsfx: %u gate(s) of chained launches gave up waiting although no launch was held up by it: the device runs this "
                                 "process's kernels one at a time and out of queue order (a profiler collecting counters?).  Calls of this batch "
                                 "stay in stream order from here on (OALSFX_DEBUG_FLAGS=0x400 does that from the start).\n", gates);
        }
        return true;
    }
    if (f >= oalsfx_hip::kFaultTurn) {
        std::snprintf(b->fault_text, sizeof(b->fault_text), "Internal error: a chained launch gave up waiting (fault word 0x%x: %u turns, %u gates, %u instances not steady).",
                      f, (f / oalsfx_hip::kFaultTurn) & 0xFFFu, f / oalsfx_hip::kFaultGate + gates, f & 0xFFFu);
        // (the instances concerned were left alone, and so were the launches behind them in the run: the batch's state is a buffer short
        // there.  No further call pretends otherwise.)
        b->poisoned = true;
        return b->fail(b->fault_text);
    }
    return b->fail("Internal error: a reverb instance listed as proven steady was not; its buffer was left unprocessed.");
}

// What a slot launches for a chunk of n frames: how many instances on the ring-light kernels, on the steady-state reverb kernels, reverbs
// in all, and which launch shape the steady ones take.
struct SlotPlan {
    int light, steady, reverbs;
    bool use_steady, mixed, by_kind, proven_usable;
    bool ragged_proven; // a call that ends in a partial tile, every steady reverb of the slot proven and at rest for its last block: the FP RG builds
};

SlotPlan plan_slot(const oalsfx_batch* b, const KernelCtx& ctx, int s, int n, bool null_has_duty)
{
    SlotPlan p{};
    for (int t = null_has_duty ? 0 : 1; t < OALSFX_REVERB; ++t) p.light += b->list_count[s][t];
    // the kernels of a slot work on disjoint instances: ring-light effects, the steady-state reverb kernel and the general
    // reverb kernel
    p.use_steady = steady_kernel_usable(ctx);
    p.steady = p.use_steady ? b->fast_count[s] + b->slow_count[s] : 0;
    p.reverbs = b->list_count[s][OALSFX_REVERB] + b->list_count[s][OALSFX_EAX_REVERB];
    // ring-light effects and steady reverbs in the same slot (mono / stereo): one grid serves both
    p.mixed = p.light > 0 && p.steady > 0 && b->channels <= 2 && !ctx.timeline && !(debug_flags() & 0x80000);
    // (the last block of a call is its shortest: up to 256 frames; the proven instances' gains are at rest for blocks of rest_tiles tiles
    // and longer)
    const int last_block = n - ((n - 1) / OALSFX_RV_MAX_UPDATE) * OALSFX_RV_MAX_UPDATE;
    const bool gains_rest = last_block >= 64 * b->rest_tiles[s];
    // (round 3: the proven instances no longer wait for the last believed one of their slot: one grid serves every kind,
    // k_reverb_steady_kinds, each workgroup on the build its instances need)
    p.by_kind = p.use_steady && !p.mixed && b->channels <= 2 && (n & 63) == 0 && !ctx.timeline;
    p.proven_usable = gains_rest && !(debug_flags() & 0x200000);
    p.ragged_proven = p.use_steady && !p.mixed && b->channels <= 2 && (n & 63) != 0 && !ctx.timeline && p.proven_usable && b->slow_count[s] == 0 &&
                      b->fast_count[s] > 0 && !(debug_flags() & 0x2000);
    return p;
}

// Ends a run of chained launches: everything queued on the batch's stream from here on comes behind the launch that may still run on
// the second stream.
bool chain_join(oalsfx_batch* b)
{
    if (!b->chain_open) return true;
    b->chain_open = false;
    for (int k = 1; k < kChainDepth; ++k) {
        if (!b->chain_used[k]) continue;
        b->chain_used[k] = false;
        if (!b->hip_ok(hipEventRecord(b->ev_chain[k], b->chain_stream[k]), "hipEventRecord") ||
            !b->hip_ok(hipStreamWaitEvent(b->stream, b->ev_chain[k], 0), "hipStreamWaitEvent")) return false;
    }
    return true;
}

// Is this process run under a tool that collects hardware counters per kernel (rocprofv3 --pmc, or a counter file)?  Such a tool runs
// one kernel at a time, and not in the order the queues were fed: the gate in front of a chained launch then waits for a launch the tool
// holds back until the gate has finished -- every gate counts out (1.3 s each; measured: bench.py under `rocprofv3 --pmc SQ_WAVES`
// failed with "a chained launch gave up waiting (... 31 gates ...)", profiles/r04k_under_the_profiler/).  Kernels that cannot overlap
// gain nothing from chaining anyway: stream order there.  (Tracing -- --kernel-trace, --stats -- does not serialise and chains as usual.)
bool kernels_serialised_by_a_tool()
{
    static const bool yes = [] {
        const char* on = std::getenv("ROCPROF_COUNTER_COLLECTION");   // rocprofv3 --pmc / -i: "1"
        const char* which = std::getenv("ROCPROF_COUNTERS");          // ... and the counters asked for
        const char* v1 = std::getenv("ROCP_METRICS");                 // rocprof (v1) with an input file
        if (std::getenv("OALSFX_IGNORE_TOOLS")) return false;         // (tests of what happens under a tool this does not recognise)
        return (on && std::atoi(on) != 0) || (which && *which) || (v1 && *v1);
    }();
    return yes;
}

// Can this call be a run of chained launches?  The batch's own stream, nobody holding its handle, no per-launch events, one chunk, and a
// step of one of the shapes whose kernels take turns: one steady-state reverb launch (whole tiles, or the proven ragged builds); the
// reverb-free slots' launch followed by the reverbs' (several slots); one grid of ring-light effects and proven reverbs.  (Any number of
// workgroups: the gate in front of a launch sees to it that all but a few workgroups of the launch before have started, however many
// rounds of the chip that launch takes -- 32 768 instances, eight rounds: 380 -> 360 us per step.)
bool chain_eligible(oalsfx_batch* b, int frames, const float* src, const float* dst, hipStream_t stream, bool uploading)
{
    if (kernels_serialised_by_a_tool() || b->chain_given_up) return false;
    if (b->chain_open && b->chain_dsts.size() >= 256) {
        // (a caller that hands in a fresh output buffer with every call: the list of a run's output buffers starts over with a new run)
        const char* lo = reinterpret_cast<const char*>(dst);
        bool known = false;
        for (const auto& d : b->chain_dsts) known |= d.first <= lo && lo < d.second;
        if (!known) return false;
    }
    // an input that is the output of a call of the current run (a feedback loop through the caller's buffers): stream order for this one
    {
        const char* lo = reinterpret_cast<const char*>(src);
        const char* hi = lo + static_cast<size_t>(b->n) * frames * b->channels * sizeof(float);
        if (b->chain_open)
            for (const auto& d : b->chain_dsts)
                if (lo < d.second && d.first < hi) return false;
    }
    if (stream != b->stream || b->stream_handed_out || (debug_flags() & (0x400 | 8)) || b->timing_every > 0 || b->d_timeline) return false;
    if (!b->uncached) return false;
    for (const auto& kv : b->pools)
        if (kv.first % 32 != 0) return false; // (a slab of delay lines ends where its last cache line ends: reverb.hip, chained launches)
    if (frames > OALSFX_MAX_CHUNK) return false;
    if (b->channels > 2) {
        // More than two channels (round 4, late): one launch of the believing build for every instance, all of them proven and at rest (no
        // general kernel behind it); its output frames written through two channels a store.  Quad / 5.1 / 7.1, 4096 EAX reverbs:
        // 63.6 -> 59.5, 69.9 -> 63.0, 81.2 -> 74.1 us per step (profiles/r04m_multichannel_chained/; round 3 had measured a loss, with one
        // write-through store per channel).  0x40000: such batches in stream order as before.
        // (6.1, seven channels a frame: one write-through store per channel made it 74.6 -> 86.5 us per step chained -- what round 3 saw; its
        // pairs now start where the frame's parity puts an even float, three pairs and one channel alone: 75.3-76.7 -> 73.1-73.2.)
        if (debug_flags() & 0x40000) return false;
        const int last_block = frames - ((frames - 1) / OALSFX_RV_MAX_UPDATE) * OALSFX_RV_MAX_UPDATE;
        return b->slots == 1 && !uploading && (frames & 63) == 0 && b->n_filtered == 0 && b->general_count[0] == 0 && b->slow_count[0] == 0 &&
               b->fast_count[0] == b->n && last_block / 64 >= b->rest_tiles[0] && !(debug_flags() & 0x200000);
    }
    if (b->slots > 1) {
        // A step of two launches (round 4): every slot but the last free of reverbs for every instance -- one launch of the ring-light
        // kernel, a wavefront per instance walking its slots -- and the last slot one grid of steady-state reverbs.  Both take turns by
        // the same word per instance, the reverb slot's: launch after launch, whichever kernel it runs.  Whole tiles, no send filters,
        // and nothing to upload (a call that has a change to put in place goes in stream order, and ends the run).
        const int last = b->slots - 1, run = reverb_free_run(b, 0);
        if (uploading || (frames & 63) != 0 || b->n_filtered > 0 || run < last || (debug_flags() & (0x8000000 | 0x40000))) return false;
        // (batches that leave workgroup places free lose by it: 1024 instances 65.5 against 61.8 us per step, 2048: 77.5 against 70.0,
        // 3072 level, 4096: 88.7 against 94.3, 6144: 134.8 against 162.2, 8192: 186.4 against 193.7 --
        // profiles/r04g_two_launch_steps/config3_by_size.txt)
        if ((b->n + 3) / 4 < 1024 && !(debug_flags() & 0x8000)) return false;
        // (no reverb anywhere: the ring-light kernel's launch would be the whole step.  Measured slower chained -- 4096 x chorus -> flanger
        // -> echo 47.4 against 44.7 us per step; see the single-slot case below -- unless the test switch asks for it)
        if (run == b->slots) return (debug_flags() & 0x80) != 0;
        if (b->fast_count[last] + b->slow_count[last] != b->n || b->general_count[last] != 0) return false;
        KernelCtx ctx{};
        ctx.frames = frames;
        const SlotPlan pl = plan_slot(b, ctx, last, frames, true);
        return pl.by_kind && !pl.mixed && pl.steady == b->n;
    }
    // Short calls of a batch that leaves workgroup slots free gain nothing from the overlap and can lose by it: the next launch's
    // workgroups are on the chip at once, waiting, beside the ones they wait for (2048 instances x 64 frames: 16.7 us per step chained,
    // 13.4 in stream order; x 128: 22.3 against 19.5; from 256 frames on, and with every slot taken, chained is level or ahead:
    // profiles/r04e_round4_end/chained/instances_and_call_sizes.txt).
    if ((b->n + 3) / 4 < 1024 && frames < 256 && !(debug_flags() & 0x8000)) return false;
    const int steady = b->fast_count[0] + b->slow_count[0];
    if (b->general_count[0] != 0) return false;
    if (steady != b->n) {
        // A slot of ring-light effects, or of ring-light effects and proven reverbs (BASELINE configs[3]): the step is one grid as well --
        // k_wave_effects with its segments, or k_slot_mixed on its proven build -- whose ring-light wavefronts take turns like the reverb
        // groups do.  Whole tiles, no send filters, nothing to upload.
        if (uploading || (frames & 63) != 0 || b->n_filtered > 0 || (debug_flags() & 0x40000)) return false;
        KernelCtx ctx{};
        ctx.frames = frames;
        const SlotPlan pl = plan_slot(b, ctx, 0, frames, true);
        if (pl.light + pl.steady != b->n || pl.reverbs != pl.steady) return false;
        // Ring-light effects alone: measured slower chained than in stream order (4096 instances, 256-frame calls: chorus 18.9 against
        // 15.1 us per step, dedicated 19.9 against 10.4, eight types in one slot 38 against 30.7 -- launches of 10 to 30 us are about what
        // the host and the gate in front of each cost; profiles/r04h_mixed_grid_chained/ring_light_chained.txt).  With reverbs in the grid
        // a launch is 45 us and more: BASELINE configs[3] 59.7 -> 51.5 us per step.  (0x80: the test switch that chains them anyway.)
        if (pl.steady == 0) return (debug_flags() & 0x80) != 0;
        // ... and what chaining gains there is the partly filled last round of the chip: a grid that fits the chip at once loses (4096
        // instances 42.8 against 36.0 us per step, 2048: 37.3 against 35.1 -- whatever the order of its workgroups), one of one to two
        // rounds gains (6144: 41.4 against 44.9; 8192: 51.6 against 59.9), four full rounds are level (16 384: 108.2 against 107.3;
        // profiles/r04h_mixed_grid_chained/config4_by_size.txt).
        if ((b->n + 3) / 4 <= 1024 && !(debug_flags() & 0x8000)) return false;
        return pl.mixed && mixed_grid_proven(b, 0, frames);
    }
    if ((frames & 63) != 0) {
        // a call that ends in a partial tile chains when its step is one launch of the proven ragged builds (no general path inside, no
        // send-filter pre-pass in front)
        KernelCtx ctx{};
        ctx.frames = frames;
        if (!plan_slot(b, ctx, 0, frames, true).ragged_proven || b->n_filtered > 0) return false;
    }
    if (b->n_filtered > 0) {
        // send filters: only when every filtered instance's build has them inside (no pre-pass launch in front of the reverb's)
        KernelCtx ctx{};
        ctx.frames = frames;
        const SlotPlan p0 = plan_slot(b, ctx, 0, frames, true);
        if (!p0.by_kind || p0.steady != b->n) return false;
        int counts[4];
        steady_kind_counts(b, 0, p0.proven_usable, counts);
        if (filters_inside_count(b, 0, counts) != b->n_filtered) return false;
    }
    return true;
}

// The next launch of a run of chained launches: its stream, the number it waits for and the one it leaves, the gate in front of it.
bool chain_next_launch(oalsfx_batch* b, KernelCtx& ctx, int depth, PendingUpload* upload, hipStream_t* stream_out)
{
    b->chain_pos_before = b->chain_open ? b->chain_pos_last : -1;
    b->chain_pos_last = b->chain_open ? b->chain_pos : -1;
    b->chain_pos = b->chain_open ? (b->chain_pos + 1) % depth : 0;
    hipStream_t stream = b->chain_stream[b->chain_pos];
    const bool first_on_its_stream = b->chain_open && !b->chain_used[b->chain_pos]; // (of the current run)
    b->chain_used[b->chain_pos] = true;
    const size_t total = static_cast<size_t>(b->n) * b->slots;
    ctx.turn = b->d_turn;
    ctx.turn_cu = b->d_turn + total + 16;
    ctx.turn_cu2 = ctx.turn_cu + total;
    ctx.turn_slot = b->slots - 1;
    // (this launch sits behind the launch two before it in its stream -- two streams taking turns -- or that launch may still run)
    ctx.turn_two_back = (b->chain_pos_before >= 0 && b->chain_pos_before != b->chain_pos) ? 1u : 0u;
    ctx.turn_wait = b->chain_open ? b->turn_counter : 0u;
    if (++b->turn_counter == 0u) b->turn_counter = 1u;
    ctx.turn_set = b->turn_counter;
    // A launch whose workgroups wait for the launch before must not take the chip before that launch has its workgroups on it: they
    // would wait for workgroups that cannot start.  Stream order does not see to that (this launch comes behind the launch two
    // before it, on its own stream; the launch before it sits in another queue, which the hardware may get to later -- the first
    // launch ever on the second stream waited for its queue to be set up while the third launch of the run filled the chip).  So
    // every workgroup of a chained launch counts itself in as it starts, and every launch but a run's first comes behind a gate
    // (one wavefront, k_chain_gate) that waits until all but a few workgroups of the launch before have.  Then a workgroup that
    // waits always waits for one that is on the chip or through: at most those few are not, fewer than the chip has places.
    unsigned* started = b->d_turn + total;
    if (!b->chain_open) {
        if (!b->hip_ok(hipEventRecord(b->ev_chain_start, b->stream), "hipEventRecord")) return false;
        b->chain_len = 1;
    } else {
        // (the first launch of the run on one of the other streams: not before the run's first could start either)
        ++b->chain_len;
        if (first_on_its_stream && !b->hip_ok(hipStreamWaitEvent(stream, b->ev_chain_start, 0), "hipStreamWaitEvent")) return false;
        // (0x800: the gate in front of a run's second launch only, as first built -- the negative control of
        // tests/test_gpu_chained.py::test_the_first_run_of_a_fresh_process)
        if (b->chain_len == 2 || !(debug_flags() & 0x800)) {
            const unsigned target = b->started_total - static_cast<uint32_t>(std::min(8, (b->n + 3) / 4 - 1));
            if (upload && upload->st) { upload->jobs.gate_started = started; upload->jobs.gate_target = target + b->gate_skew; } // (the upload kernel is the gate as well)
            else oalsfx_hip::launch_chain_gate(started, target + b->gate_skew, b->d_fault + 1, stream);
        }
    }
    ctx.turn_started = started;
    b->chain_open = true;
    b->launched_groups = 0;
    *stream_out = stream;
    return true;
}

// ... and behind it: every workgroup of the launch counts itself in on the device (reverb.hip, turn_started): the grid as launched, which
// for a grid of several kinds is up to three workgroups more than a quarter of the instances (ADVICE, round 3: the host added (n + 3) / 4
// and drifted behind the device's count by up to three per call).
bool chain_launch_done(oalsfx_batch* b)
{
    if (b->launched_groups <= 0) return b->fail("Internal error: a chained call launched no grid that takes turns.");
    b->started_total += static_cast<uint32_t>(b->launched_groups);
    b->launched_groups = 0;
    return true;
}

bool mix_device(oalsfx_batch* b, int frames, const float* src, float* dst, hipStream_t stream, bool may_chain = false)
{
    if (b->poisoned) return b->fail(b->fault_text);
    poll_exact(b);
    b->launched_groups = 0;
    // what has changed since the last call: the host's part first (it decides what this call launches), the upload itself below, where
    // the call's launches go
    PendingUpload upload;
    const auto hp0 = std::chrono::steady_clock::now();
    if (!prepare_params(b, upload)) return false;
    b->host_prepare_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - hp0).count();
    const bool chained = may_chain && (!upload.st || upload.chainable) && chain_eligible(b, frames, src, dst, stream, upload.st != nullptr);
    if (!chained && !chain_join(b)) return false;
    int depth = 2;
    if (chained) {
        // The streams in turn; the first launch of a run stays on the batch's stream, behind whatever was queued there before.  Two streams
        // while every workgroup takes the same time (one kind of proven instances, nothing uploaded), three when workgroups differ
        // (several kinds, cross-fading instances): a launch then does not wait for the slowest workgroups of the launch two before it.
        const int rs = b->slots - 1; // (the reverbs' slot)
        int populated = 0;
        for (int k = 0; k < 3; ++k) populated += b->kind_count[rs][k] > 0;
        // (Round 4 looked at three for the uniform workload again: the gate in front of a launch is a kernel of its own behind the launch
        // two before it, and with two streams it starts when that launch ends, 5 us before the launch behind it can be dispatched
        // (profiles/r04f_round4_end/chained/timeline_chained.txt); with three it is through by then.  400-call runs 40.3 -> 39.8 us per
        // step -- and the driver's own command, bench.py --steps 20 --warmup 5, 43.0-43.7 -> 49.0-51.6: a short run pays for the third
        // stream's start three times over.  Two it stays; OALSFX_DEBUG_FLAGS 0x10000: three.
        // profiles/r04c_instruction_diet/chain_depth_uniform.txt)
        depth = (populated > 1 || b->slow_count[rs] > 0 || upload.st || (debug_flags() & 0x10000)) ? kChainDepth : std::min(2, kChainDepth);
        static const int forced_depth = std::getenv("OALSFX_CHAIN_DEPTH") ? std::atoi(std::getenv("OALSFX_CHAIN_DEPTH")) : 0; // (experiments)
        if (forced_depth >= 2 && forced_depth <= kChainDepth) depth = forced_depth;
    }
    if (!chained && !launch_params(b, upload, b->stream, stream)) return false;
    if (!ensure_mixbuf(b)) return false;
    b->timing = b->timing_every > 0 && (b->mix_calls++ % b->timing_every) == 0;
    const bool filtered = b->n_filtered > 0;
    const int chunk_max = std::min(frames, OALSFX_MAX_CHUNK);
    const size_t plane = static_cast<size_t>(b->n) * chunk_max * b->channels;
    if (filtered && plane > b->filtered_capacity) {
        if (!b->hip_ok(hipStreamSynchronize(stream), "hipStreamSynchronize")) return false;
        for (int k = 1; k < kChainDepth && b->chain_open; ++k) // (a run of chained launches: the planes' readers may be on its other streams)
            if (!b->hip_ok(hipStreamSynchronize(b->chain_stream[k]), "hipStreamSynchronize")) return false;
        hipFree(b->d_filtered);
        b->d_filtered = nullptr;
        b->filtered_capacity = 0;
        if (!b->hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_filtered), plane * (1 + b->slots) * sizeof(float)), "hipMalloc(filtered sends)")) return false;
        b->filtered_capacity = plane;
    }
    KernelCtx ctx{};
    ctx.params = b->d_params;
    ctx.state = b->d_state;
    ctx.rings = b->d_rings;
    ctx.source = b->d_source;
    ctx.source_state = b->d_source_state;
    ctx.mixbuf = b->d_mixbuf;
    ctx.timeline = b->d_timeline;
    ctx.hot = b->d_hot;
    ctx.inst_epoch = b->d_inst_epoch;
    ctx.exact = b->d_exact;
    ctx.fault = b->d_fault;
    ctx.list_first = -1;
    ctx.no_follow_up = 0;
    if (chained) {
        {
            const char* lo = reinterpret_cast<const char*>(dst);
            const char* hi = lo + static_cast<size_t>(b->n) * frames * b->channels * sizeof(float);
            if (!b->chain_open) b->chain_dsts.clear();
            bool known = false;
            for (auto& d : b->chain_dsts) known |= d.first <= lo && hi <= d.second;
            if (!known) b->chain_dsts.push_back({lo, hi});
        }
        // the step's first launch (of two, for a batch of several slots: chain_eligible)
        if (!chain_next_launch(b, ctx, depth, &upload, &stream)) return false;
        // Parameters that changed since the call before: put in place on this launch's stream, behind the gate -- beside the launch
        // before, which may still be at work with the old ones: a slot's record (and its instance's epoch) is stored once that launch is
        // through with the instance; a rebuilt list went to the buffer that launch does not read.
        if (!launch_params(b, upload, stream, nullptr, b->d_turn, ctx.turn_wait)) return false;
        b->chained_calls += 1;
    }
    ctx.slots = b->slots;
    ctx.channels = b->channels;
    ctx.io_stride = static_cast<long long>(frames) * b->channels;
    // (a chained step of two kernels: workgroups of one size, so that either kernel's fit the places the other's give up -- common.hpp)
    struct EqualPlaces {
        bool on;
        explicit EqualPlaces(bool o) : on(o)
        {
            static const int bytes = std::getenv("OALSFX_EQUAL_LDS") ? std::atoi(std::getenv("OALSFX_EQUAL_LDS")) : 40960; // (experiments)
            if (on) oalsfx_hip::set_lds_per_workgroup(bytes);
        }
        ~EqualPlaces() { if (on) oalsfx_hip::set_lds_per_workgroup(0); }
    } equal_places(chained && b->slots > 1 && !(debug_flags() & 0x1000));
    // Api::mix chunking (reference src/oalsfxpp.cpp:3818-3826)
    for (int done = 0; done < frames;) {
        const int n = std::min(frames - done, OALSFX_MAX_CHUNK);
        const float* chunk_src = src + static_cast<size_t>(done) * b->channels;
        ctx.raw_src = chunk_src;
        ctx.dst = dst + static_cast<size_t>(done) * b->channels;
        ctx.frames = n;
        bool planes = filtered; // the pre-pass ran: the instances it covers read their sends from its planes (flag kFiltered)
        // Single-slot batches: the steady-state reverb builds of the first two kinds filter their own sends (SF); the pre-pass then covers
        // the rest of the slot's list only, or is not launched at all.
        bool filters_inside = false;
        if (filtered && b->slots == 1) {
            const SlotPlan p0 = plan_slot(b, ctx, 0, n, true);
            if (p0.by_kind && p0.steady > 0) {
                int counts[4];
                steady_kind_counts(b, 0, p0.proven_usable, counts);
                const int inside = filters_inside_count(b, 0, counts);
                if (inside > 0) {
                    filters_inside = true;
                    ctx.src_stride = static_cast<long long>(n) * b->channels;
                    planes = inside < b->n_filtered;
                    if (planes) {
                        const int before = b->steady_offset[0], behind = before + counts[0] + counts[1];
                        oalsfx_hip::launch_send_filters(ctx, chunk_src, ctx.io_stride, b->d_filtered, b->filtered_capacity, b->d_lists, before, stream);
                        oalsfx_hip::launch_send_filters(ctx, chunk_src, ctx.io_stride, b->d_filtered, b->filtered_capacity, b->d_lists + behind, b->n - behind, stream);
                    }
                }
            }
        }
        if (filters_inside) {
            ctx.src = planes ? b->d_filtered : chunk_src;
            if (!planes) ctx.src_stride = ctx.io_stride;
        } else if (filtered) {
            // apply_filters for every send (reference src/oalsfxpp.cpp:2929-2965): planes of [instance][n][channels]
            ctx.src_stride = static_cast<long long>(n) * b->channels;
            oalsfx_hip::launch_send_filters(ctx, chunk_src, ctx.io_stride, b->d_filtered, b->filtered_capacity, nullptr, b->n, stream);
            ctx.src = b->d_filtered;
        } else {
            ctx.src_stride = ctx.io_stride;
            ctx.src = chunk_src;
        }
        ctx.wet_plane = planes ? static_cast<long long>(b->filtered_capacity) : 0;
        for (int s = 0; s < b->slots; ++s) {
            ctx.wet_src = planes ? b->d_filtered + static_cast<size_t>(1 + s) * b->filtered_capacity : ctx.src;
            const int run = reverb_free_run(b, s);
            if (run >= 2 && !(debug_flags() & 0x8000000)) {
                // slots s .. s+run-1 hold ring-light effects (or nothing) for every instance: one launch, one wavefront per
                // instance, the slots in order inside it; the slot's list in type order lists every instance exactly once
                const int run_flags = (s == 0 ? oalsfx_hip::kFirst : 0) | (s + run == b->slots ? oalsfx_hip::kLast : 0) |
                                      (planes ? oalsfx_hip::kFiltered : 0);
                {
                    ScopedTiming timing(b, kTimedWaveEffects, stream);
                    // (a chained step: in the order of the reverbs' grid -- its list names every instance once as well)
                    const bool grid_order = chained && s + run < b->slots && !(debug_flags() & 0x10);
                    const int* every = grid_order ? b->d_lists + b->steady_offset[b->slots - 1] : b->d_lists + b->list_offset[s][OALSFX_NULL];
                    oalsfx_hip::launch_wave_effects(ctx, s, run, every, b->n, nullptr,
                                                    run_flags | (chained ? ((debug_flags() & 7) | (grid_order ? 0 : 8)) << 8 : 0), stream);
                }
                if (chained) b->launched_groups = (b->n + 3) / 4;
                s += run - 1;
                continue;
            }
            // (a chained step's second launch: the reverbs' grid, next in the run)
            if (chained && s > 0 && (!chain_launch_done(b) || !chain_next_launch(b, ctx, depth, nullptr, &stream))) return false;
            const int flags = (s == 0 ? oalsfx_hip::kFirst : 0) | (s == b->slots - 1 ? oalsfx_hip::kLast : 0) |
                              (planes ? oalsfx_hip::kFiltered : 0);
            const bool null_has_duty = (flags & (oalsfx_hip::kFirst | oalsfx_hip::kLast)) != 0;
            const SlotPlan sp = plan_slot(b, ctx, s, n, null_has_duty);
            const int light = sp.light, steady = sp.steady, reverbs = sp.reverbs;
            const bool use_steady = sp.use_steady, mixed = sp.mixed, by_kind = sp.by_kind, proven_usable = sp.proven_usable;
            // parts: ring-light effects | (unused) | steady reverbs | general reverbs
            bool part_on[4] = {light > 0 && !mixed, false, steady > 0, reverbs - steady > 0};
            int parts = 0;
            for (bool on : part_on) parts += on;
            const bool fork = parts > 1 && !(debug_flags() & 0x20000);
            if (fork && !b->hip_ok(hipEventRecord(b->ev_fork, stream), "hipEventRecord")) return false;
            // The caller's stream takes the part the step will wait for longest -- the general kernel, if there is one (a few
            // wavefronts, each a long chain of latencies) -- and the others go beside it: what the stream then waits for at the
            // join has long finished, and neither fork nor join (about 12 us each on this stack) lies on the step's critical path.
            int main_part = -1;
            for (int g = 0; g < 4; ++g)
                if (part_on[g]) main_part = (main_part < 0 || g == 3) ? g : main_part;
            if (debug_flags() & 0x20000000) { main_part = -1; for (int g = 3; g >= 0; --g) if (part_on[g]) main_part = g; } // experiment: the first part, as before
            int side = 0;
            for (int g = 0; g < 4; ++g) {
                if (!part_on[g]) continue;
                hipStream_t gs = stream;
                if (fork && g != main_part) {
                    gs = b->side_stream[side];
                    if (!b->hip_ok(hipStreamWaitEvent(gs, b->ev_fork, 0), "hipStreamWaitEvent")) return false;
                }
                if (g == 0) {
                    launch_wave_group(b, ctx, s, flags, gs);
                } else if (g == 2 && mixed) {
                    launch_mixed_part(b, ctx, s, flags, gs);
                } else if (g == 2 && by_kind) {
                    launch_reverb_kinds_part(b, ctx, s, flags, proven_usable, filters_inside, gs);
                } else if (g == 2) {
                    // ragged calls, more than two channels, the timeline build: one launch of the believing builds for all of them -- or,
                    // for a ragged call of proven instances only, of the proven ragged builds
                    launch_reverb_steady_part(b, ctx, s, flags, b->steady_offset[s], steady, sp.ragged_proven, gs);
                } else {
                    launch_reverb_general_part(b, !use_steady, ctx, s, flags, gs);
                }
                if (gs != stream) {
                    if (!b->hip_ok(hipEventRecord(b->ev_join[side], gs), "hipEventRecord")) return false;
                    ++side;
                }
            }
            for (int k = 0; k < side; ++k)
                if (!b->hip_ok(hipStreamWaitEvent(stream, b->ev_join[k], 0), "hipStreamWaitEvent")) return false;
        }
        done += n;
    }
    if (chained && !chain_launch_done(b)) return false;
    if ((frames % OALSFX_MAX_CHUNK) & 63) {
        // a ragged chunk: a cross-fade in flight no longer stands at a tile boundary, which the XF build needs
        for (size_t idx : b->settling)
            if (b->xf_ok[idx]) { b->xf_ok[idx] = 0; b->lists_dirty = true; }
    }
    advance_settling(b, frames);
    b->frames_total += static_cast<uint32_t>(frames);
    if (frames & 31) b->off_grid_known = false;
    if (b->exact_wanted && !b->exact_pending) {
        // what this call's kernels found out about the reverbs that are not proven steady yet, fetched behind them
        const size_t total = static_cast<size_t>(b->n) * b->slots;
        if (!b->hip_ok(hipMemcpyAsync(b->h_exact, b->d_exact, total * sizeof(unsigned), hipMemcpyDeviceToHost, stream), "hipMemcpyAsync(exact)")) return false;
        if (!b->hip_ok(hipEventRecord(b->ev_exact, stream), "hipEventRecord")) return false;
        b->exact_pending = true;
        b->exact_wanted = false;
        b->exact_gen = b->upload_gen;
    }
    b->last_launch_stream = chained ? b->stream : stream;
    if (!chained && stream != b->stream && !b->hip_ok(hipEventRecord(b->ev_mixed, stream), "hipEventRecord")) return false;
    return b->hip_ok(hipGetLastError(), "kernel launch");
}

} // namespace

extern "C" {

const char* oalsfx_last_error(void) { return g_last_error.c_str(); }

oalsfx_batch* oalsfx_batch_create(int n_instances, int channel_format, int sampling_rate, int effect_count, int device_id)
{
    g_last_error.clear();
    const int channels = channel_count_of(static_cast<oalsfxpp::ChannelFormat>(channel_format));
    // same checks and messages as Api::Impl::initialize (reference src/oalsfxpp.cpp:2853-2871)
    if (channels == 0) { g_last_error = "Invalid channel format."; return nullptr; }
    if (sampling_rate < min_sampling_rate) { g_last_error = "Sampling rate is out of range."; return nullptr; }
    if (effect_count <= 0 || effect_count > OALSFX_MAX_SLOTS) { g_last_error = "Effect count is out of range."; return nullptr; }
    if (n_instances <= 0) { g_last_error = "Instance count is out of range."; return nullptr; }

    int device_count = 0;
    if (hipGetDeviceCount(&device_count) != hipSuccess || device_count <= 0) {
        g_last_error = "No HIP device available: the effect process path has no CPU fallback.";
        return nullptr;
    }
    if (device_id < 0 || device_id >= device_count) { g_last_error = "HIP device ordinal is out of range."; return nullptr; }
    if (hipSetDevice(device_id) != hipSuccess) { g_last_error = "hipSetDevice failed."; return nullptr; }

    auto* b = new (std::nothrow) oalsfx_batch{};
    if (!b) { g_last_error = "Failed to allocate the batch."; return nullptr; }
    b->n = n_instances;
    b->slots = effect_count;
    b->channels = channels;
    b->rate = sampling_rate;
    b->format = channel_format;
    b->device = device_id;
    b->dev.init(static_cast<oalsfxpp::ChannelFormat>(channel_format), sampling_rate);
    b->channels = b->dev.channels;
    const size_t total = static_cast<size_t>(n_instances) * effect_count;
    b->inst.resize(n_instances);
    b->h_params.assign(total, oalsfx_slot_params{});
    b->h_state_init.assign(total, oalsfx_slot_state{});
    b->started_at.assign(total, 0);
    b->h_source.assign(n_instances, oalsfx_source_params{});
    b->seq.assign(total, 0);
    b->h_rings.assign(total, nullptr);
    b->ring_floats.assign(total, 0);
    b->inst_dirty.assign(n_instances, 0);
    b->inst_filtered.assign(n_instances, 0);
    b->touched.assign(n_instances, 0);
    b->since_update.assign(total, 0);
    b->slot_class.assign(total, 0);
    b->in_settling.assign(total, 0);
    b->xf_ok.assign(total, 0);
    b->mod_ever.assign(total, 0);
    b->proven.assign(total, 0);
    b->updated_gen.assign(total, 0);
    b->inst_epoch.assign(n_instances, 1); // a zero-filled hot record never carries a valid stamp
    for (int i = 0; i < n_instances; ++i) {
        b->inst[i].initialize(effect_count);
        mark_dirty(b, i);
    }
    mark_touched(b, 0, n_instances);

    bool ok = b->hip_ok(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking), "hipStreamCreate");
    b->chain_stream[0] = b->stream;
    for (int k = 1; k < kChainDepth; ++k) {
        ok = ok && b->hip_ok(hipStreamCreateWithFlags(&b->chain_stream[k], hipStreamNonBlocking), "hipStreamCreate");
        ok = ok && b->hip_ok(hipEventCreateWithFlags(&b->ev_chain[k], hipEventDisableTiming), "hipEventCreate");
    }
    {
        // can a call of this batch ever be a chained launch?  (chain_eligible has the conditions that change from call to call)
        const char* kind = std::getenv("OALSFX_RING_MEMORY");
        b->uncached = (!kind || std::strcmp(kind, "uncached") == 0) && uncached_memory_available(b->device);
    }
    ok = ok && b->hip_ok(hipEventCreateWithFlags(&b->ev_chain_start, hipEventDisableTiming), "hipEventCreate");
    for (int k = 0; k < kSideStreams; ++k) {
        ok = ok && b->hip_ok(hipStreamCreateWithFlags(&b->side_stream[k], hipStreamNonBlocking), "hipStreamCreate");
        ok = ok && b->hip_ok(hipEventCreateWithFlags(&b->ev_join[k], hipEventDisableTiming), "hipEventCreate");
    }
    ok = ok && b->hip_ok(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming), "hipEventCreate");
    ok = ok && b->hip_ok(hipEventCreateWithFlags(&b->ev_uploaded, hipEventDisableTiming), "hipEventCreate");
    ok = ok && b->hip_ok(hipEventCreateWithFlags(&b->ev_mixed, hipEventDisableTiming), "hipEventCreate");
    ok = ok && b->hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_params), total * sizeof(oalsfx_slot_params)), "hipMalloc(params)");
    ok = ok && b->hip_ok(handed_on_malloc(b, reinterpret_cast<void**>(&b->d_state), total * sizeof(oalsfx_hip::SlotStateLines)), "hipMalloc(state)");
    ok = ok && b->hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_source), n_instances * sizeof(oalsfx_source_params)), "hipMalloc(source)");
    ok = ok && b->hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_rings), total * sizeof(float*)), "hipMalloc(ring table)");
    ok = ok && b->hip_ok(handed_on_malloc(b, reinterpret_cast<void**>(&b->d_source_state), n_instances * sizeof(oalsfx_source_state)), "hipMalloc(source state)");
    ok = ok && b->hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_lists_buf[0]), (kChainDepth + 1) * total * sizeof(int)), "hipMalloc(lists)");
    if (ok) {
        for (int k = 1; k <= kChainDepth; ++k) b->d_lists_buf[k] = b->d_lists_buf[0] + k * total;
        b->d_lists = b->d_lists_buf[0];
    }
    ok = ok && b->hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_progress), total * sizeof(int)), "hipMalloc(progress)");
    ok = ok && b->hip_ok(hipMemsetAsync(b->d_progress, 0, total * sizeof(int), b->stream), "hipMemsetAsync(progress)");
    ok = ok && b->hip_ok(handed_on_malloc(b, reinterpret_cast<void**>(&b->d_hot), total * oalsfx_hip::hot::SIZE * sizeof(unsigned)), "hipMalloc(hot records)");
    ok = ok && b->hip_ok(hipMemsetAsync(b->d_hot, 0, total * oalsfx_hip::hot::SIZE * sizeof(unsigned), b->stream), "hipMemsetAsync(hot records)");
    ok = ok && b->hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_inst_epoch), n_instances * sizeof(unsigned)), "hipMalloc(epochs)");
    ok = ok && b->hip_ok(hipMemsetAsync(b->d_inst_epoch, 0, n_instances * sizeof(unsigned), b->stream), "hipMemsetAsync(epochs)");
    ok = ok && b->hip_ok(handed_on_malloc(b, reinterpret_cast<void**>(&b->d_turn), (3 * total + 16) * sizeof(unsigned)), "hipMalloc(turns)");
    ok = ok && b->hip_ok(hipMemsetAsync(b->d_turn, 0, (3 * total + 16) * sizeof(unsigned), b->stream), "hipMemsetAsync(turns)");
    ok = ok && b->hip_ok(handed_on_malloc(b, reinterpret_cast<void**>(&b->d_exact), total * sizeof(unsigned)), "hipMalloc(exact)");
    ok = ok && b->hip_ok(hipMemsetAsync(b->d_exact, 0, total * sizeof(unsigned), b->stream), "hipMemsetAsync(exact)");
    ok = ok && b->hip_ok(hipHostMalloc(reinterpret_cast<void**>(&b->h_exact), total * sizeof(unsigned)), "hipHostMalloc(exact)");
    ok = ok && b->hip_ok(hipHostMalloc(reinterpret_cast<void**>(&b->h_fault), 2 * sizeof(unsigned), hipHostMallocMapped), "hipHostMalloc(fault)");
    if (ok) b->h_fault[0] = b->h_fault[1] = 0;
    ok = ok && b->hip_ok(hipHostGetDevicePointer(reinterpret_cast<void**>(&b->d_fault), b->h_fault, 0), "hipHostGetDevicePointer");
    ok = ok && b->hip_ok(hipEventCreateWithFlags(&b->ev_exact, hipEventDisableTiming), "hipEventCreate");
    ok = ok && b->hip_ok(hipMemsetAsync(b->d_state, 0, total * sizeof(oalsfx_hip::SlotStateLines), b->stream), "hipMemsetAsync(state)");
    ok = ok && b->hip_ok(hipMemsetAsync(b->d_source_state, 0, n_instances * sizeof(oalsfx_source_state), b->stream), "hipMemsetAsync(source state)");
    if (ok && std::getenv("OALSFX_DEBUG_TIMELINE")) {
        ok = b->hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_timeline), kTimelineBytes), "hipMalloc(timeline)");
        ok = ok && b->hip_ok(hipMemsetAsync(b->d_timeline, 0, kTimelineBytes, b->stream), "hipMemsetAsync(timeline)");
    }
    if (!ok) {
        g_last_error = b->error;
        oalsfx_batch_destroy(b);
        return nullptr;
    }
    return b;
}

void oalsfx_batch_destroy(oalsfx_batch* b)
{
    if (!b) return;
    hipSetDevice(b->device);
    for (int k = 1; k < kChainDepth; ++k)
        if (b->chain_stream[k]) hipStreamSynchronize(b->chain_stream[k]);
    if (b->stream) hipStreamSynchronize(b->stream);
    for (int k = 0; k < kSideStreams; ++k)
        if (b->side_stream[k]) hipStreamSynchronize(b->side_stream[k]);
    if (b->exact_pending) hipEventSynchronize(b->ev_exact); // a read-back on a caller's stream still writes to h_exact
    if (b->d_timeline) {
        // phase stamps of the last steady-state reverb launch, for scripts/timeline.py
        std::vector<unsigned long long> h(kTimelineBytes / sizeof(unsigned long long));
        if (hipMemcpy(h.data(), b->d_timeline, kTimelineBytes, hipMemcpyDeviceToHost) == hipSuccess)
            if (FILE* f = std::fopen(std::getenv("OALSFX_DEBUG_TIMELINE"), "wb")) { std::fwrite(h.data(), 1, kTimelineBytes, f); std::fclose(f); }
        hipFree(b->d_timeline);
    }
    for (auto& t : b->timed) { hipEventDestroy(t.start); hipEventDestroy(t.stop); }
    for (hipEvent_t e : b->event_pool) hipEventDestroy(e);
    if (std::getenv("OALSFX_HOST_PROFILE"))
        std::fprintf(stderr, "host profile: prepare_params %.1f ms (of which waiting for a staging buffer %.1f ms), %lld mix_device calls chained\n",
                     b->host_prepare_ns * 1e-6, b->host_stage_wait_ns * 1e-6, b->chained_calls);
    for (void* c : b->chunks) handed_on_free(c);
    hipFree(b->d_params); handed_on_free(b->d_state); hipFree(b->d_source); hipFree(b->d_rings); handed_on_free(b->d_source_state); hipFree(b->d_filtered);
    handed_on_free(b->d_mixbuf); hipFree(b->d_lists_buf[0]); hipFree(b->d_progress); hipFree(b->d_io_src); hipFree(b->d_io_dst);
    handed_on_free(b->d_hot); hipFree(b->d_inst_epoch); handed_on_free(b->d_exact); handed_on_free(b->d_turn);
    for (int k = 1; k < kChainDepth; ++k)
        if (b->ev_chain[k]) hipEventDestroy(b->ev_chain[k]);
    if (b->ev_chain_start) hipEventDestroy(b->ev_chain_start);
    for (int k = 1; k < kChainDepth; ++k)
        if (b->chain_stream[k]) hipStreamDestroy(b->chain_stream[k]);
    if (b->h2d_stream) hipStreamSynchronize(b->h2d_stream);
    if (b->d2h_stream) hipStreamSynchronize(b->d2h_stream);
    for (auto& ps : b->pipe) {
        hipFree(ps.d_src); hipFree(ps.d_dst);
        if (ps.copied_in) hipEventDestroy(ps.copied_in);
        if (ps.mixed) hipEventDestroy(ps.mixed);
        if (ps.copied_out) hipEventDestroy(ps.copied_out);
    }
    if (b->h2d_stream) hipStreamDestroy(b->h2d_stream);
    if (b->d2h_stream) hipStreamDestroy(b->d2h_stream);
    if (b->h_io_src) (void)hipHostFree(b->h_io_src);
    if (b->h_io_dst) (void)hipHostFree(b->h_io_dst);
    if (b->h_exact) (void)hipHostFree(b->h_exact);
    if (b->h_fault) (void)hipHostFree(b->h_fault);
    if (b->ev_exact) hipEventDestroy(b->ev_exact);
    for (int k = 0; k < kSideStreams; ++k) {
        if (b->side_stream[k]) hipStreamDestroy(b->side_stream[k]);
        if (b->ev_join[k]) hipEventDestroy(b->ev_join[k]);
    }
    if (b->ev_fork) hipEventDestroy(b->ev_fork);
    if (b->ev_uploaded) hipEventDestroy(b->ev_uploaded);
    if (b->ev_mixed) hipEventDestroy(b->ev_mixed);
    for (auto& st : b->stage) {
        if (st.host) (void)hipHostFree(st.host);
        if (st.dev) (void)hipFree(st.dev);
        if (st.done) hipEventDestroy(st.done);
    }
    if (b->stream) hipStreamDestroy(b->stream);
    delete b;
}

const char* oalsfx_batch_error(const oalsfx_batch* b) { return b ? b->error : g_last_error.c_str(); }
int oalsfx_batch_instances(const oalsfx_batch* b) { return b->n; }
int oalsfx_batch_channels(const oalsfx_batch* b) { return b->channels; }
int oalsfx_batch_sampling_rate(const oalsfx_batch* b) { return b->rate; }
int oalsfx_batch_effect_count(const oalsfx_batch* b) { return b->slots; }

int oalsfx_batch_set_effect(oalsfx_batch* b, int first, int count, int slot, const oalsfx_effect* effects, int stride_bytes)
{
    if (!range_ok(b, first, count)) return 0;
    if (slot < 0 || slot >= b->slots) return b->fail(kErrSlot) ? 1 : 0;
    const auto* base = reinterpret_cast<const unsigned char*>(effects);
    for (int i = 0; i < count; ++i)
        std::memcpy(&b->inst[first + i].deferred[slot], base + static_cast<size_t>(i) * stride_bytes, sizeof(oalsfx_effect));
    mark_touched(b, first, count);
    return 1;
}

int oalsfx_batch_set_effect_at(oalsfx_batch* b, const int* instances, int count, int slot, const oalsfx_effect* effects, int stride_bytes)
{
    if (count < 0 || (count > 0 && (!instances || !effects))) return b->fail(kErrRange) ? 1 : 0;
    if (slot < 0 || slot >= b->slots) return b->fail(kErrSlot) ? 1 : 0;
    for (int k = 0; k < count; ++k)
        if (instances[k] < 0 || instances[k] >= b->n) return b->fail(kErrRange) ? 1 : 0; // (nothing is set when one index is out of range)
    const auto* base = reinterpret_cast<const unsigned char*>(effects);
    for (int k = 0; k < count; ++k) {
        std::memcpy(&b->inst[instances[k]].deferred[slot], base + static_cast<size_t>(k) * stride_bytes, sizeof(oalsfx_effect));
        mark_touched(b, instances[k], 1);
    }
    return 1;
}

int oalsfx_batch_set_effect_type(oalsfx_batch* b, int first, int count, int slot, int effect_type)
{
    if (!range_ok(b, first, count)) return 0;
    if (slot < 0 || slot >= b->slots) return b->fail(kErrSlot) ? 1 : 0;
    for (int i = 0; i < count; ++i) b->inst[first + i].deferred[slot].set_type_and_defaults(static_cast<oalsfxpp::EffectType>(effect_type));
    mark_touched(b, first, count);
    return 1;
}

int oalsfx_batch_set_effect_props(oalsfx_batch* b, int first, int count, int slot, const void* props, int stride_bytes)
{
    if (!range_ok(b, first, count)) return 0;
    if (slot < 0 || slot >= b->slots) return b->fail(kErrSlot) ? 1 : 0;
    const auto* base = static_cast<const unsigned char*>(props);
    for (int i = 0; i < count; ++i)
        std::memcpy(&b->inst[first + i].deferred[slot].props_, base + static_cast<size_t>(i) * stride_bytes, sizeof(oalsfxpp::EffectProps));
    mark_touched(b, first, count);
    return 1;
}

int oalsfx_batch_set_send_props(oalsfx_batch* b, int first, int count, int slot, const oalsfx_send_props* props)
{
    if (!range_ok(b, first, count)) return 0;
    if (slot >= b->slots) return b->fail(kErrSlot) ? 1 : 0;
    oalsfxpp::SendProps p;
    p.gain_ = props->gain; p.gain_hf_ = props->gain_hf; p.gain_lf_ = props->gain_lf;
    for (int i = 0; i < count; ++i) {
        InstanceHost& h = b->inst[first + i];
        // direct: deferred copy; auxiliary: the reference writes the active properties (src/oalsfxpp.cpp:3728-3733)
        if (slot < 0) h.direct_deferred = p;
        else h.aux_props[slot] = p;
    }
    mark_touched(b, first, count);
    return 1;
}

int oalsfx_batch_get_effect(const oalsfx_batch* b, int instance, int slot, int deferred, oalsfx_effect* out)
{
    if (instance < 0 || instance >= b->n || slot < 0 || slot >= b->slots) return 0;
    const InstanceHost& h = b->inst[instance];
    std::memcpy(out, deferred ? &h.deferred[slot] : &h.active[slot], sizeof(oalsfx_effect));
    return 1;
}

int oalsfx_batch_get_send_props(const oalsfx_batch* b, int instance, int slot, int deferred, oalsfx_send_props* out)
{
    if (instance < 0 || instance >= b->n || slot >= b->slots) return 0;
    const InstanceHost& h = b->inst[instance];
    const oalsfxpp::SendProps& p = slot < 0 ? (deferred ? h.direct_deferred : h.direct_props) : (deferred ? h.aux_deferred[slot] : h.aux_props[slot]);
    out->gain = p.gain_; out->gain_hf = p.gain_hf_; out->gain_lf = p.gain_lf_;
    return 1;
}

int oalsfx_batch_apply_changes(oalsfx_batch* b, int first, int count)
{
    if (!range_ok(b, first, count)) return 0;
    size_t kept = 0;
    for (size_t k = 0; k < b->touched_list.size(); ++k) {
        const int i = b->touched_list[k];
        if (i < first || i >= first + count) { b->touched_list[kept++] = i; continue; }
        InstanceHost& h = b->inst[i];
        h.apply_changes();
        bool dirty = h.source_changed;
        for (int s = 0; s < b->slots; ++s) dirty |= h.slot_changed[s];
        if (dirty) mark_dirty(b, i);
        // An auxiliary send whose properties were ever set differs from its never-written deferred copy for good, and the
        // reference then recomputes that source on every apply (src/oalsfxpp.cpp:3772-3780): such an instance stays listed.
        bool sticky = false;
        for (int s = 0; s < b->slots; ++s) sticky |= !oalsfxpp::SendProps::are_equal(h.aux_props[s], h.aux_deferred[s]);
        if (sticky) b->touched_list[kept++] = i;
        else b->touched[i] = 0;
    }
    b->touched_list.resize(kept);
    return 1;
}

int oalsfx_batch_mix_device(oalsfx_batch* b, int frames, const float* src_dev, float* dst_dev, void* hip_stream)
{
    if (frames == 0) return 1;
    if (frames < 0) return b->fail("Frame count is negative.") ? 1 : 0;
    if (!src_dev) return b->fail(kErrNoSrc) ? 1 : 0;
    if (!dst_dev) return b->fail(kErrNoDst) ? 1 : 0;
    if (!b->hip_ok(hipSetDevice(b->device), "hipSetDevice")) return 0;
    // (calls on the batch's own stream may overlap with their neighbours: chained launches)
    return mix_device(b, frames, src_dev, dst_dev, hip_stream ? static_cast<hipStream_t>(hip_stream) : b->stream, hip_stream == nullptr) ? 1 : 0;
}

namespace {

// Api::mix from host buffers: copy in, kernels, copy out on the batch's stream, then wait.  legs_us (may be null): the three legs as
// HIP events on that stream saw them.
int mix_host(oalsfx_batch* b, int frames, const float* src_host, float* dst_host, double* legs_us)
{
    if (legs_us) legs_us[0] = legs_us[1] = legs_us[2] = 0.0;
    if (frames == 0) return 1;
    if (frames < 0) return b->fail("Frame count is negative.") ? 1 : 0;
    if (!src_host) return b->fail(kErrNoSrc) ? 1 : 0;
    if (!dst_host) return b->fail(kErrNoDst) ? 1 : 0;
    if (!b->hip_ok(hipSetDevice(b->device), "hipSetDevice") || !chain_join(b)) return 0;
    const size_t floats = static_cast<size_t>(b->n) * frames * b->channels;
    if (floats > b->io_capacity) {
        hipFree(b->d_io_src); hipFree(b->d_io_dst);
        b->d_io_src = b->d_io_dst = nullptr;
        b->io_capacity = 0;
        if (!b->hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_io_src), floats * sizeof(float)), "hipMalloc(io)")) return 0;
        if (!b->hip_ok(hipMalloc(reinterpret_cast<void**>(&b->d_io_dst), floats * sizeof(float)), "hipMalloc(io)")) return 0;
        b->io_capacity = floats;
    }
    hipEvent_t ev[4] = {};
    if (legs_us)
        for (auto& e : ev) e = b->take_event();
    if (legs_us) hipEventRecord(ev[0], b->stream);
    if (!b->hip_ok(hipMemcpyAsync(b->d_io_src, src_host, floats * sizeof(float), hipMemcpyHostToDevice, b->stream), "hipMemcpyAsync(src)")) return 0;
    if (legs_us) hipEventRecord(ev[1], b->stream);
    if (!mix_device(b, frames, b->d_io_src, b->d_io_dst, b->stream)) return 0;
    if (legs_us) hipEventRecord(ev[2], b->stream);
    if (!b->hip_ok(hipMemcpyAsync(dst_host, b->d_io_dst, floats * sizeof(float), hipMemcpyDeviceToHost, b->stream), "hipMemcpyAsync(dst)")) return 0;
    if (legs_us) hipEventRecord(ev[3], b->stream);
    if (!b->hip_ok(hipStreamSynchronize(b->stream), "hipStreamSynchronize")) return 0;
    if (legs_us) {
        for (int k = 0; k < 3; ++k) {
            float ms = 0.0F;
            if (hipEventElapsedTime(&ms, ev[k], ev[k + 1]) == hipSuccess) legs_us[k] = ms * 1e3;
        }
        for (auto& e : ev) b->event_pool.push_back(e);
    }
    poll_exact(b);
    return check_fault(b) ? 1 : 0;
}

} // namespace

int oalsfx_batch_mix(oalsfx_batch* b, int frames, const float* src_host, float* dst_host) { return mix_host(b, frames, src_host, dst_host, nullptr); }

// Api::mix for callers that keep one source and one target buffer per instance -- thousands of oalsfxpp::Api objects' worth of them
// (reference src/oalsfxpp.h:872-875 takes one pair per object): the frames are gathered into one page-locked buffer, go through the
// batch as one call, and are scattered back.  The two passes over host memory cost more than the device's part (8 MB each way at 4096
// stereo instances of 256 frames), and still some thirty times less than a batch of one per object.
int oalsfx_batch_mix_gather(oalsfx_batch* b, int frames, const float* const* src_per_instance, float* const* dst_per_instance)
{
    if (frames == 0) return 1;
    if (frames < 0) return b->fail("Frame count is negative.") ? 1 : 0;
    if (!src_per_instance) return b->fail(kErrNoSrc) ? 1 : 0;
    if (!dst_per_instance) return b->fail(kErrNoDst) ? 1 : 0;
    const size_t per = static_cast<size_t>(frames) * b->channels, floats = per * b->n;
    for (int i = 0; i < b->n; ++i) {
        if (!src_per_instance[i]) return b->fail(kErrNoSrc) ? 1 : 0;
        if (!dst_per_instance[i]) return b->fail(kErrNoDst) ? 1 : 0;
    }
    if (!b->hip_ok(hipSetDevice(b->device), "hipSetDevice")) return 0;
    if (floats > b->h_io_capacity) {
        if (b->h_io_src) (void)hipHostFree(b->h_io_src);
        if (b->h_io_dst) (void)hipHostFree(b->h_io_dst);
        b->h_io_src = b->h_io_dst = nullptr;
        b->h_io_capacity = 0;
        if (!b->hip_ok(hipHostMalloc(reinterpret_cast<void**>(&b->h_io_src), floats * sizeof(float), hipHostMallocDefault), "hipHostMalloc(io)")) return 0;
        if (!b->hip_ok(hipHostMalloc(reinterpret_cast<void**>(&b->h_io_dst), floats * sizeof(float), hipHostMallocDefault), "hipHostMalloc(io)")) return 0;
        b->h_io_capacity = floats;
    }
    for (int i = 0; i < b->n; ++i) std::memcpy(b->h_io_src + per * i, src_per_instance[i], per * sizeof(float));
    if (!mix_host(b, frames, b->h_io_src, b->h_io_dst, nullptr)) return 0;
    for (int i = 0; i < b->n; ++i) std::memcpy(dst_per_instance[i], b->h_io_dst + per * i, per * sizeof(float));
    return 1;
}

int oalsfx_batch_mix_timed(oalsfx_batch* b, int frames, const float* src_host, float* dst_host, double legs_us[3])
{
    return mix_host(b, frames, src_host, dst_host, legs_us);
}

namespace {

std::mutex g_pipe_mutex;
std::map<int, int> g_pipe_mode; // device -> form, once a batch has probed it

// Which form this call of oalsfx_batch_mix_async takes (oalsfx_batch::pipe_mode); while probing, also the probe's bookkeeping.
int pipe_form(oalsfx_batch* b)
{
    if (b->pipe_mode) return b->pipe_mode;
    if (const char* e = std::getenv("OALSFX_HOST_PIPELINE")) {
        const int m = std::atoi(e);
        if (m == 1 || m == 3) return b->pipe_mode = m;
    }
    {
        std::lock_guard<std::mutex> lock(g_pipe_mutex);
        auto it = g_pipe_mode.find(b->device);
        if (it != g_pipe_mode.end()) return b->pipe_mode = it->second;
    }
    // probing: calls 0 .. 15 on three streams (streams, events and staging buffers come into being, the copy engines see the caller's
    // buffers for the first time: not timed -- these calls took 1.2 ms each where the steady state takes 0.24), 16 .. 31 on one stream,
    // 32 .. 47 on three again; the clock runs over the last twelve of the second and third sixteen
    const long long k = b->pipe_turn;
    const auto now = std::chrono::steady_clock::now();
    if ((k & 15) == 4) b->pipe_probe_t0 = now;
    if (k == 32 || k == 48) b->pipe_probe_us[k == 48 ? 0 : 1] = std::chrono::duration<double, std::micro>(now - b->pipe_probe_t0).count() / 12.0;
    if (k < 48) return (k >= 16 && k < 32) ? 1 : 3;
    b->pipe_mode = b->pipe_probe_us[1] < b->pipe_probe_us[0] * 0.9 ? 1 : 3;
    std::lock_guard<std::mutex> lock(g_pipe_mutex);
    g_pipe_mode[b->device] = b->pipe_mode;
    return b->pipe_mode;
}

} // namespace

int oalsfx_batch_mix_async(oalsfx_batch* b, int frames, const float* src_host, float* dst_host)
{
    if (frames == 0) return 1;
    if (frames < 0) return b->fail("Frame count is negative.") ? 1 : 0;
    if (!src_host) return b->fail(kErrNoSrc) ? 1 : 0;
    if (!dst_host) return b->fail(kErrNoDst) ? 1 : 0;
    if (!b->hip_ok(hipSetDevice(b->device), "hipSetDevice") || !chain_join(b)) return 0;
    const size_t floats = static_cast<size_t>(b->n) * frames * b->channels;
    if (!b->h2d_stream) {
        if (!b->hip_ok(hipStreamCreateWithFlags(&b->h2d_stream, hipStreamNonBlocking), "hipStreamCreate")) return 0;
        if (!b->hip_ok(hipStreamCreateWithFlags(&b->d2h_stream, hipStreamNonBlocking), "hipStreamCreate")) return 0;
        for (auto& ps : b->pipe) {
            if (!b->hip_ok(hipEventCreateWithFlags(&ps.copied_in, hipEventDisableTiming), "hipEventCreate")) return 0;
            if (!b->hip_ok(hipEventCreateWithFlags(&ps.mixed, hipEventDisableTiming), "hipEventCreate")) return 0;
            if (!b->hip_ok(hipEventCreateWithFlags(&ps.copied_out, hipEventDisableTiming), "hipEventCreate")) return 0;
        }
    }
    if (floats > b->pipe_capacity) {
        // a larger call than any before: let the pipeline drain, then grow every slot
        if (!b->hip_ok(hipStreamSynchronize(b->d2h_stream), "hipStreamSynchronize") || !b->hip_ok(hipStreamSynchronize(b->stream), "hipStreamSynchronize")) return 0;
        for (auto& ps : b->pipe) {
            hipFree(ps.d_src); hipFree(ps.d_dst);
            ps.d_src = ps.d_dst = nullptr;
            ps.busy = false;
        }
        b->pipe_capacity = 0;
        for (auto& ps : b->pipe) {
            if (!b->hip_ok(hipMalloc(reinterpret_cast<void**>(&ps.d_src), floats * sizeof(float)), "hipMalloc(io)")) return 0;
            if (!b->hip_ok(hipMalloc(reinterpret_cast<void**>(&ps.d_dst), floats * sizeof(float)), "hipMalloc(io)")) return 0;
        }
        b->pipe_capacity = floats;
    }
    const int form = pipe_form(b);
    oalsfx_batch::PipeSlot& ps = b->pipe[b->pipe_turn++ % oalsfx_batch::kPipeDepth];
    // the call that used this staging slot kPipeDepth calls ago must be through: its output copy is the last thing it does
    if (ps.busy && !b->hip_ok(hipEventSynchronize(ps.copied_out), "hipEventSynchronize")) return 0;
    ps.busy = true;
    if (form == 1) {
        // one stream: the synchronous call's sequence, without its wait
        if (!b->hip_ok(hipMemcpyAsync(ps.d_src, src_host, floats * sizeof(float), hipMemcpyHostToDevice, b->stream), "hipMemcpyAsync(src)")) return 0;
        if (!mix_device(b, frames, ps.d_src, ps.d_dst, b->stream)) return 0;
        if (!b->hip_ok(hipMemcpyAsync(dst_host, ps.d_dst, floats * sizeof(float), hipMemcpyDeviceToHost, b->stream), "hipMemcpyAsync(dst)")) return 0;
        return b->hip_ok(hipEventRecord(ps.copied_out, b->stream), "hipEventRecord") ? 1 : 0;
    }
    // The copies go through the runtime's copy engines.  (Experiment, OALSFX_DEBUG_FLAGS 0x800000: as kernels reading / writing the
    // page-locked buffers directly.  Measured slower, 0.32 against 0.20 ms per step: the reverb grid holds every CU, and the copy
    // kernels' workgroups wait for its slots.)
    bool by_kernel = false;
    if (debug_flags() & 0x800000) {
        auto mapped = [](const void* p) {
            hipPointerAttribute_t a{};
            return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeHost && a.devicePointer != nullptr;
        };
        by_kernel = mapped(src_host) && mapped(dst_host);
        (void)hipGetLastError(); // hipPointerGetAttributes on pageable memory leaves an error behind
    }
    if (by_kernel) {
        oalsfx_hip::launch_copy_floats(ps.d_src, src_host, floats, b->h2d_stream);
    } else if (!b->hip_ok(hipMemcpyAsync(ps.d_src, src_host, floats * sizeof(float), hipMemcpyHostToDevice, b->h2d_stream), "hipMemcpyAsync(src)")) {
        return 0;
    }
    if (!b->hip_ok(hipEventRecord(ps.copied_in, b->h2d_stream), "hipEventRecord")) return 0;
    if (!b->hip_ok(hipStreamWaitEvent(b->stream, ps.copied_in, 0), "hipStreamWaitEvent")) return 0;
    if (!mix_device(b, frames, ps.d_src, ps.d_dst, b->stream)) return 0;
    if (!b->hip_ok(hipEventRecord(ps.mixed, b->stream), "hipEventRecord")) return 0;
    if (!b->hip_ok(hipStreamWaitEvent(b->d2h_stream, ps.mixed, 0), "hipStreamWaitEvent")) return 0;
    if (by_kernel) {
        oalsfx_hip::launch_copy_floats(dst_host, ps.d_dst, floats, b->d2h_stream);
    } else if (!b->hip_ok(hipMemcpyAsync(dst_host, ps.d_dst, floats * sizeof(float), hipMemcpyDeviceToHost, b->d2h_stream), "hipMemcpyAsync(dst)")) {
        return 0;
    }
    return b->hip_ok(hipEventRecord(ps.copied_out, b->d2h_stream), "hipEventRecord") ? 1 : 0;
}

int oalsfx_batch_wait(oalsfx_batch* b)
{
    if (!b->hip_ok(hipSetDevice(b->device), "hipSetDevice") || !chain_join(b)) return 0;
    if (b->d2h_stream && !b->hip_ok(hipStreamSynchronize(b->d2h_stream), "hipStreamSynchronize")) return 0;
    if (!b->hip_ok(hipStreamSynchronize(b->stream), "hipStreamSynchronize")) return 0;
    for (auto& ps : b->pipe) ps.busy = false;
    poll_exact(b);
    return check_fault(b) ? 1 : 0;
}

void* oalsfx_pinned_alloc(unsigned long long bytes)
{
    void* p = nullptr;
    return hipHostMalloc(&p, bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}

void oalsfx_pinned_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

int oalsfx_batch_synchronize(oalsfx_batch* b)
{
    if (!b->hip_ok(hipSetDevice(b->device), "hipSetDevice") || !chain_join(b)) return 0;
    if (!b->hip_ok(hipStreamSynchronize(b->stream), "hipStreamSynchronize")) return 0;
    if (std::getenv("OALSFX_HOST_PROFILE")) {
        std::fprintf(stderr, "host profile since the last synchronize: prepare_params %.1f us, of which staging-buffer waits %.1f us, derive %.1f us, lists %.1f us; %lld calls chained so far\n",
                     b->host_prepare_ns * 1e-3, b->host_stage_wait_ns * 1e-3, b->host_derive_ns * 1e-3, b->host_lists_ns * 1e-3, b->chained_calls);
        b->host_prepare_ns = b->host_stage_wait_ns = b->host_derive_ns = b->host_lists_ns = 0;
    }
    poll_exact(b);
    return check_fault(b) ? 1 : 0;
}

void* oalsfx_batch_stream(oalsfx_batch* b)
{
    // whoever holds the handle may queue work that expects the launches in stream order: no more chained launches for this batch
    b->stream_handed_out = true;
    if (hipSetDevice(b->device) == hipSuccess) (void)chain_join(b);
    return b->stream;
}

int oalsfx_batch_read_slot(oalsfx_batch* b, int instance, int slot, oalsfx_slot_params* params, oalsfx_slot_state* state)
{
    if (instance < 0 || instance >= b->n || slot < 0 || slot >= b->slots) return b->fail(kErrRange) ? 1 : 0;
    if (!b->hip_ok(hipSetDevice(b->device), "hipSetDevice") || !chain_join(b)) return 0;
    if (!sync_params(b, nullptr)) return 0;
    const size_t idx = static_cast<size_t>(instance) * b->slots + slot;
    if (params) *params = b->h_params[idx];
    if (state) {
        if (!b->hip_ok(hipStreamSynchronize(b->stream), "hipStreamSynchronize")) return 0;
        if (!b->hip_ok(hipMemcpy(state, static_cast<const oalsfx_slot_state*>(b->d_state + idx), sizeof(*state), hipMemcpyDeviceToHost), "hipMemcpy(state)")) return 0;
    }
    return 1;
}

int oalsfx_batch_read_ring(oalsfx_batch* b, int instance, int slot, float* out, int max_floats)
{
    if (instance < 0 || instance >= b->n || slot < 0 || slot >= b->slots) return 0;
    if (hipSetDevice(b->device) != hipSuccess || !chain_join(b) || !sync_params(b, nullptr)) return 0;
    const size_t idx = static_cast<size_t>(instance) * b->slots + slot;
    const int floats = static_cast<int>(b->ring_floats[idx]);
    if (out && floats) {
        hipStreamSynchronize(b->stream);
        const int n = std::min(floats, max_floats);
        if (!b->hip_ok(hipMemcpy(out, b->h_rings[idx], static_cast<size_t>(n) * sizeof(float), hipMemcpyDeviceToHost), "hipMemcpy(ring)")) return 0;
    }
    return floats;
}

int oalsfx_batch_read_source(oalsfx_batch* b, int instance, oalsfx_source_params* params, oalsfx_source_state* state)
{
    if (instance < 0 || instance >= b->n) return b->fail(kErrRange) ? 1 : 0;
    if (!b->hip_ok(hipSetDevice(b->device), "hipSetDevice") || !chain_join(b)) return 0;
    if (!sync_params(b, nullptr)) return 0;
    if (params) *params = b->h_source[instance];
    if (state) {
        if (!b->hip_ok(hipStreamSynchronize(b->stream), "hipStreamSynchronize")) return 0;
        if (!b->hip_ok(hipMemcpy(state, b->d_source_state + instance, sizeof(*state), hipMemcpyDeviceToHost), "hipMemcpy(source state)")) return 0;
    }
    return 1;
}

int oalsfx_batch_fill_synthetic(oalsfx_batch* b, int frames, unsigned buffer_index, float* dst_dev, void* hip_stream)
{
    if (!b->hip_ok(hipSetDevice(b->device), "hipSetDevice") || !chain_join(b)) return 0;
    oalsfx_hip::launch_fill_synthetic(dst_dev, b->n, frames * b->channels, buffer_index, hip_stream ? static_cast<hipStream_t>(hip_stream) : b->stream);
    return b->hip_ok(hipGetLastError(), "fill_synthetic") ? 1 : 0;
}

int oalsfx_batch_kernel_timing(oalsfx_batch* b, int enable)
{
    if (!b->hip_ok(hipSetDevice(b->device), "hipSetDevice") || !chain_join(b)) return 0;
    for (auto& t : b->timed) { b->event_pool.push_back(t.start); b->event_pool.push_back(t.stop); }
    b->timed.clear();
    b->timing_every = enable > 0 ? enable : 0;
    b->timing = false;
    b->mix_calls = 0;
    while (b->timing_every && b->event_pool.size() < 2048) {
        hipEvent_t e = nullptr;
        if (!b->hip_ok(hipEventCreate(&e), "hipEventCreate")) return 0;
        b->event_pool.push_back(e);
    }
    return 1;
}

int oalsfx_batch_kernel_timing_read(oalsfx_batch* b, int effect_type, int* launches, double* total_ms)
{
    if (!b->hip_ok(hipSetDevice(b->device), "hipSetDevice") || !chain_join(b)) return 0;
    int n = 0;
    double ms = 0.0;
    // the ring-light types share one launch per slot: asking for any of them reads that launch
    int key = (effect_type >= 0 && effect_type < OALSFX_REVERB) ? kTimedWaveEffects : effect_type;
    // the two reverb types share their launches
    if (key == OALSFX_REVERB) key = OALSFX_EAX_REVERB;
    if (key == OALSFX_EAX_REVERB + kTimedGeneralOffset) key = OALSFX_REVERB + kTimedGeneralOffset;
    if (effect_type == 32) key = kTimedMixed;
    for (auto& t : b->timed) {
        if (t.type != key) continue;
        if (!b->hip_ok(hipEventSynchronize(t.stop), "hipEventSynchronize")) return 0;
        float e = 0.0F;
        if (!b->hip_ok(hipEventElapsedTime(&e, t.start, t.stop), "hipEventElapsedTime")) return 0;
        ms += e;
        ++n;
    }
    if (launches) *launches = n;
    if (total_ms) *total_ms = ms;
    return 1;
}

int oalsfx_batch_kernel_timing_samples(oalsfx_batch* b, int effect_type, double* out_us, int max_samples)
{
    if (!b->hip_ok(hipSetDevice(b->device), "hipSetDevice") || !chain_join(b)) return -1;
    int key = (effect_type >= 0 && effect_type < OALSFX_REVERB) ? kTimedWaveEffects : effect_type;
    if (key == OALSFX_REVERB) key = OALSFX_EAX_REVERB;
    if (key == OALSFX_EAX_REVERB + kTimedGeneralOffset) key = OALSFX_REVERB + kTimedGeneralOffset;
    if (effect_type == 32) key = kTimedMixed;
    int n = 0;
    for (auto& t : b->timed) {
        if (t.type != key) continue;
        if (out_us && n < max_samples) {
            if (!b->hip_ok(hipEventSynchronize(t.stop), "hipEventSynchronize")) return -1;
            float e = 0.0F;
            if (!b->hip_ok(hipEventElapsedTime(&e, t.start, t.stop), "hipEventElapsedTime")) return -1;
            out_us[n] = e * 1e3;
        }
        ++n;
    }
    return n;
}

int oalsfx_batch_plan(oalsfx_batch* b, int slot, int counts[4])
{
    if (slot < 0 || slot >= b->slots) return b->fail(kErrSlot) ? 1 : 0;
    if (!b->hip_ok(hipSetDevice(b->device), "hipSetDevice") || !chain_join(b)) return 0;
    poll_exact(b);
    if (!sync_params(b, nullptr)) return 0;
    int light = 0;
    for (int t = 0; t < OALSFX_REVERB; ++t) light += b->list_count[slot][t];
    counts[0] = light;
    counts[1] = b->fast_count[slot];
    counts[2] = b->slow_count[slot];
    counts[3] = b->general_count[slot];
    return 1;
}

const char* oalsfx_batch_last_reverb_kernel(const oalsfx_batch* b) { return b->last_steady_kernel; }

long long oalsfx_batch_chained_calls(const oalsfx_batch* b) { return b->chained_calls; }

long long oalsfx_debug_chain_same_cu(oalsfx_batch* b)
{
    // how many instance hand-overs of chained launches stayed on one CU (and paid for an L1 invalidate) so far
    unsigned v = 0;
    if (hipSetDevice(b->device) != hipSuccess || !chain_join(b) || hipStreamSynchronize(b->stream) != hipSuccess) return -1;
    if (hipMemcpy(&v, b->d_turn + static_cast<size_t>(b->n) * b->slots + 1, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return v;
}

int oalsfx_debug_host_pipeline(oalsfx_batch* b, int* form, double* probe_us_three_streams, double* probe_us_one_stream)
{
    if (form) *form = b->pipe_mode;
    if (probe_us_three_streams) *probe_us_three_streams = b->pipe_probe_us[0];
    if (probe_us_one_stream) *probe_us_one_stream = b->pipe_probe_us[1];
    return 1;
}

unsigned long long oalsfx_trim_pools(void)
{
    return uncached_pool().trim(0);
}

unsigned long long oalsfx_pools_waiting_bytes(void) { return uncached_pool().bytes_waiting(); }

void oalsfx_debug_gate_skew(oalsfx_batch* b, unsigned skew) { if (b) b->gate_skew = skew; }
int oalsfx_debug_chain_given_up(const oalsfx_batch* b) { return b && b->chain_given_up ? 1 : 0; }

int oalsfx_debug_chain_started(oalsfx_batch* b, unsigned* host_total, unsigned* device_total)
{
    unsigned v = 0;
    if (!b->hip_ok(hipSetDevice(b->device), "hipSetDevice") || !chain_join(b) || !b->hip_ok(hipStreamSynchronize(b->stream), "hipStreamSynchronize")) return 0;
    if (b->d_turn && !b->hip_ok(hipMemcpy(&v, b->d_turn + static_cast<size_t>(b->n) * b->slots, sizeof(v), hipMemcpyDeviceToHost), "hipMemcpy")) return 0;
    if (host_total) *host_total = b->started_total;
    if (device_total) *device_total = v;
    return 1;
}

int oalsfx_device_pci_bus_id(int device_id, char* out, int len)
{
    if (!out || len < 16) return 0;
    return hipDeviceGetPCIBusId(out, len, device_id) == hipSuccess ? 1 : 0;
}

int oalsfx_batch_event_overhead(oalsfx_batch* b, int repeats, double* avg_us)
{
    if (repeats <= 0 || !b->hip_ok(hipSetDevice(b->device), "hipSetDevice") || !chain_join(b)) return 0;
    hipEvent_t e0 = b->take_event(), e1 = b->take_event();
    double total = 0.0;
    bool ok = true;
    for (int r = 0; r < repeats && ok; ++r) {
        oalsfx_hip::launch_null(b->stream); // something ahead in the stream, as in the timed region
        hipEventRecord(e0, b->stream);
        hipEventRecord(e1, b->stream);
        ok = hipEventSynchronize(e1) == hipSuccess;
        float ms = 0.0F;
        ok = ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
        total += ms;
    }
    b->event_pool.push_back(e0);
    b->event_pool.push_back(e1);
    if (avg_us) *avg_us = total * 1e3 / repeats;
    return ok ? 1 : 0;
}

int oalsfx_batch_placement(const oalsfx_batch* b, int* chunks, int* candidates, double* best_us, double* worst_us)
{
    if (chunks) *chunks = b->placed_chunks;
    if (candidates) *candidates = b->placement_candidates;
    if (best_us) *best_us = b->placement_best_us;
    if (worst_us) *worst_us = b->placement_worst_us;
    return 1;
}

int oalsfx_debug_probe_pointer(void* slabs, int instances, int slab_floats, int repeats, double* avg_us)
{
    // the placement probe on any device buffer of instances * slab_floats floats, always 4096 wavefronts (scripts/vram_map.py)
    if (!slabs || instances <= 0 || repeats <= 0 || slab_floats < 65536) return 0;
    const int wps = std::max(1, 4096 / instances);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 2; ++r) oalsfx_hip::launch_ring_probe(static_cast<float*>(slabs), instances, slab_floats, 256u * r, wps, nullptr);
    hipEventRecord(e0, nullptr);
    for (int r = 0; r < repeats; ++r) oalsfx_hip::launch_ring_probe(static_cast<float*>(slabs), instances, slab_floats, 256u * (2 + r), wps, nullptr);
    hipEventRecord(e1, nullptr);
    bool ok = hipEventSynchronize(e1) == hipSuccess;
    float ms = 0.0F;
    ok = ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
    if (avg_us) *avg_us = ms * 1e3 / repeats;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ok ? 1 : 0;
}

int oalsfx_debug_probe_rings(oalsfx_batch* b, int repeats, double* avg_us)
{
    // the reverb's ring traffic without its arithmetic (k_stream_pattern) on the batch's own delay-line chunk: overwrites the delay lines
    if (hipSetDevice(b->device) != hipSuccess || !chain_join(b) || !sync_params(b, nullptr) || b->ring_chunks.empty() || repeats <= 0) return 0;
    if (!b->hip_ok(hipStreamSynchronize(b->stream), "hipStreamSynchronize")) return 0;
    const auto& rc = b->ring_chunks.front();
    const size_t slab_floats = static_cast<size_t>(ring_floats_for(OALSFX_EAX_REVERB, b->rate));
    const int instances = static_cast<int>(rc.bytes / (slab_floats * sizeof(float)));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 4; ++r) oalsfx_hip::launch_stream_pattern(reinterpret_cast<float*>(rc.base), instances, 1, 256u * r, slab_floats, 0, b->stream);
    hipEventRecord(e0, b->stream);
    for (int r = 0; r < repeats; ++r) oalsfx_hip::launch_stream_pattern(reinterpret_cast<float*>(rc.base), instances, 1, 256u * (4 + r), slab_floats, 0, b->stream);
    hipEventRecord(e1, b->stream);
    bool ok = hipEventSynchronize(e1) == hipSuccess;
    float ms = 0.0F;
    ok = ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
    if (avg_us) *avg_us = ms * 1e3 / repeats;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ok ? 1 : 0;
}

int oalsfx_debug_move_rings(oalsfx_batch* b, int keep_old)
{
    if (hipSetDevice(b->device) != hipSuccess || !chain_join(b) || !sync_params(b, nullptr)) return 0;
    if (!b->hip_ok(hipStreamSynchronize(b->stream), "hipStreamSynchronize")) return 0;
    const size_t total = static_cast<size_t>(b->n) * b->slots;
    for (auto& rc : b->ring_chunks) {
        void* fresh = nullptr;
        if (!b->hip_ok(handed_on_malloc(b, &fresh, rc.bytes), "hipMalloc(rings)")) return 0;
        if (!b->hip_ok(hipMemcpy(fresh, rc.base, rc.bytes, hipMemcpyDeviceToDevice), "hipMemcpy(rings)")) return 0;
        const ptrdiff_t delta = static_cast<char*>(fresh) - rc.base;
        auto inside = [&](float* p) { return reinterpret_cast<char*>(p) >= rc.base && reinterpret_cast<char*>(p) < rc.base + rc.bytes; };
        auto moved = [&](float* p) { return reinterpret_cast<float*>(reinterpret_cast<char*>(p) + delta); };
        for (size_t idx = 0; idx < total; ++idx)
            if (b->h_rings[idx] && inside(b->h_rings[idx])) b->h_rings[idx] = moved(b->h_rings[idx]);
        for (auto& kv : b->pools) {
            for (auto& p : kv.second.free_clean) if (inside(p)) p = moved(p);
            for (auto& p : kv.second.free_dirty) if (inside(p)) p = moved(p);
        }
        if (keep_old) {
            // stays allocated (and counted in `chunks`) so that the next move lands somewhere else again
        } else {
            for (auto& c : b->chunks) if (c == rc.alloc) c = nullptr;
            handed_on_free(rc.alloc);
        }
        b->chunks.push_back(fresh);
        rc.base = static_cast<char*>(fresh);
        rc.alloc = fresh;
    }
    return b->hip_ok(hipMemcpy(b->d_rings, b->h_rings.data(), total * sizeof(float*), hipMemcpyHostToDevice), "hipMemcpy(ring table)") ? 1 : 0;
}

unsigned long long oalsfx_debug_ring_address(oalsfx_batch* b, int instance, int slot)
{
    if (instance < 0 || instance >= b->n || slot < 0 || slot >= b->slots) return 0;
    if (hipSetDevice(b->device) != hipSuccess || !chain_join(b) || !sync_params(b, nullptr)) return 0;
    return reinterpret_cast<unsigned long long>(b->h_rings[static_cast<size_t>(instance) * b->slots + slot]);
}

void oalsfx_debug_set_flags(int flags) { g_debug_flags.store(flags < 0 ? 0 : flags, std::memory_order_relaxed); }

int oalsfx_debug_hbm_sweep(int device_id, unsigned long long bytes, int write, int repeats)
{
    if (hipSetDevice(device_id) != hipSuccess) return 0;
    float* buf = nullptr;
    float* sink = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&buf), bytes) != hipSuccess) return 0;
    if (hipMalloc(reinterpret_cast<void**>(&sink), 256) != hipSuccess) { (void)hipFree(buf); return 0; }
    (void)hipMemset(buf, 0, bytes);
    for (int r = 0; r < repeats; ++r) oalsfx_hip::launch_hbm_sweep(buf, bytes / sizeof(float), write, sink, nullptr);
    const bool ok = hipDeviceSynchronize() == hipSuccess;
    (void)hipFree(buf);
    (void)hipFree(sink);
    return ok ? 1 : 0;
}

int oalsfx_debug_stream_pattern(int device_id, int instances, int dwords_per_lane, int repeats, int slab_floats, int pos_skew, double* avg_us)
{
    if (hipSetDevice(device_id) != hipSuccess || instances <= 0 || repeats <= 0) return 0;
    if (dwords_per_lane != 1 && dwords_per_lane != 2 && dwords_per_lane != 4) return 0;
    float* slabs = nullptr;
    if (slab_floats < 235520) return 0;
    const size_t bytes = static_cast<size_t>(instances) * slab_floats * sizeof(float);
    if (hipMalloc(reinterpret_cast<void**>(&slabs), bytes) != hipSuccess) return 0;
    (void)hipMemset(slabs, 0, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 8; ++r) oalsfx_hip::launch_stream_pattern(slabs, instances, dwords_per_lane, 256u * r, slab_floats, pos_skew, nullptr);
    hipEventRecord(e0, nullptr);
    for (int r = 0; r < repeats; ++r) oalsfx_hip::launch_stream_pattern(slabs, instances, dwords_per_lane, 256u * (8 + r), slab_floats, pos_skew, nullptr);
    hipEventRecord(e1, nullptr);
    bool ok = hipEventSynchronize(e1) == hipSuccess;
    float ms = 0.0F;
    ok = ok && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
    if (avg_us) *avg_us = ms * 1e3 / repeats;
    hipEventDestroy(e0); hipEventDestroy(e1);
    (void)hipFree(slabs);
    return ok ? 1 : 0;
}

} // extern "C"
