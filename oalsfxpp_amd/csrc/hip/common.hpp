// Shared declarations of the HIP side: launch context, device helpers, launcher prototypes.
//
// Build flags that matter for parity (set in oalsfxpp_amd/build.py): -ffp-contract=off (no FMA
// contraction: the reference is evaluated with separately rounded mul/add, SURVEY 7 hard part 1),
// no fast-math, IEEE division/sqrt (hipcc default), fp32 denormals on (gfx9 default).
#ifndef OALSFX_HIP_COMMON_HPP
#define OALSFX_HIP_COMMON_HPP

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "oalsfx_desc.h"

namespace oalsfx_hip {

// A slot's state in device memory: the ABI's record in 128-byte lines of its own.  No cache line holds parts of two instances' records
// (hot records: 1 KiB each; send-filter histories: 1280 bytes each; delay lines: slabs of whole lines), which is what lets an instance
// change hands between two launches that run at the same time with no cache invalidated in between (chained launches, reverb.hip).
struct alignas(128) SlotStateLines : oalsfx_slot_state {};
static_assert(sizeof(SlotStateLines) % 128 == 0 && sizeof(oalsfx_source_state) % 128 == 0, "an instance's records end where its cache lines end");


// What every effect kernel needs to find its instance's data.  Passed by value.
struct KernelCtx {
    const oalsfx_slot_params* params;   // [instance][slots]
    SlotStateLines* state;              // [instance][slots]
    float* const* rings;                // [instance][slots] -> ring slab of that slot (or nullptr)
    const oalsfx_source_params* source; // [instance]
    oalsfx_source_state* source_state;  // [instance]: histories of the send filters
    const float* raw_src;               // [instance][frames][channels] interleaved input of this chunk (stride io_stride)
    const float* src;                   // ... as the direct send sees it: raw_src, or with kFiltered the pre-pass plane of the direct send
                                        // (stride src_stride), valid for the instances that have a send filter switched on
    const float* wet_src;               // ... as this slot's auxiliary send sees it (same rule)
    long long wet_plane;                // floats between the wet_src planes of consecutive slots (0 unless kFiltered)
    float* dst;                         // [instance][frames][channels] interleaved output of this chunk
    float* mixbuf;                      // [instance][channels][OALSFX_MAX_CHUNK] planar accumulator (multi-slot only)
    int slots;
    int channels;
    int frames;                         // frames in this chunk (<= OALSFX_MAX_CHUNK)
    long long io_stride;                // floats between consecutive instances in dst
    long long src_stride;               // floats between consecutive instances in src / wet_src
    int* progress;                      // [instance][slots]: frames of this chunk the steady-state reverb kernel has done (all it was
                                        // given, or none), written when it runs without its inside fallback (more than two
                                        // channels) and read by the general kernel launched right after on the same list, which
                                        // carries on from there; nullptr otherwise
    unsigned long long* timeline;       // measurement only (OALSFX_DEBUG_TIMELINE): phase time stamps of sampled workgroups, else nullptr
    // ---- the proven-steady reverb path (reverb.hip, build flag FP) ----
    unsigned* hot;                      // [instance][slots][hot::SIZE] dwords: what a proven-steady reverb needs to start a buffer, packed (see namespace hot)
    const unsigned* inst_epoch;         // [instance]: bumped by the host with every parameter upload that touches the instance; stamps the hot records
    unsigned* exact;                    // [instance][slots]: written by every reverb kernel except the FP builds: 1 when the instance ends the call
                                        // with its cross-fade finished and every output gain exactly on its target (then "steady" no longer
                                        // depends on the size of the next call), else 0; the host reads it back lazily
    unsigned* fault;                    // one counter: instances an FP build found not to be steady after all (a broken host invariant: reported
                                        // by the next synchronising call, never silently)
    // ---- consecutive calls that overlap (batch.cpp: chained launches) ----
    unsigned* turn;                     // [instance][slots]: the number the last steady-state launch left for the instance; nullptr: off
    unsigned turn_wait;                 // != 0: an instance starts once its word holds this number (the launch before this one, on another
                                        // stream, is through with it)
    unsigned turn_set;                  // != 0: what an instance's word is set to when the launch is through with it
    unsigned* turn_started;             // every workgroup counts itself in here as it starts (k_chain_gate)
    unsigned* turn_cu;                  // [instance][slots]: the CU the last chained launch ran the instance on (reverb.hip, this_cu)
    unsigned* turn_cu2;                 // ... and the one before it
    unsigned turn_two_back;             // != 0: the launch before the last may still have been at work when this launch started (it is not
                                        // the launch this one sits behind in its stream)
    int turn_slot;                      // a step of several launches (batch.cpp: the reverb-free slots' launch, then the reverb slot's): the slot
                                        // whose word every launch of the step takes turns by -- the reverb's own (it indexes by its slot)
    int list_first;                     // >= 0: the launch's list is the range list_first, list_first + 1, ... (no list load); -1: read the list
    int no_follow_up;        // hand-over launches (ctx.progress): the first no_follow_up entries of the list are proven steady and the general kernel
                              // will not be run on them; one that is not steady after all is counted in `fault`
};

// The packed per-(instance, slot) start record of the proven-steady reverb kernel: the LDS table image, the filter histories and the
// wave-uniform odds and ends, 1 KiB, written by the kernel's epilogue for the next call and fetched with one 16-byte load per lane.
// Valid when its stamp (host epoch of the instance, delay-line position) matches the instance's current ones; rebuilt from the
// descriptors otherwise.
namespace hot {
enum {
    UT = 0,        // 128 dwords: namespace ut of reverb.hip
    CHAIN = 128,   // 64 dwords: [line][coop::SIZE] filter histories and feedback coefficients
    MISC = 192,    // 64 dwords, see below
    SIZE = 256,
    // MISC dwords
    M_EPOCH = 0, M_OFFSET, M_EAX, M_HAS_FILTER, M_AUD_DIR, M_AUD_AUX, M_AUD_OUT, M_LATE_MASK, M_SHORT_MASK, M_MOD_F, M_MOD_INDEX, M_MOD_RANGE,
    M_MOD_DEPTH, M_MOD_COEFF, M_MOD_ON, M_SEND_MASK, M_COUNT
};
}

// How the workgroups of a ring-light grid map to a slot's type-sorted instance list: segment k covers the next `count[k]` list
// entries, four per workgroup (a segment starts a new workgroup).  A segment whose bit is set in coop_mask holds whole
// workgroups of one effect type, which run their filter recurrences together (wave_effects_body.hpp, chain_phase).
// n == 0: no segments, wavefront w takes list entry w.
// The segments are given in the order the grid's workgroups take them, which need not be the list's (an experiment switch puts the
// effect types whose workgroups run longest first; list order measured faster); offset[k] says where segment k starts in the list.
struct WaveSegments {
    enum { kMax = 20 }; // ten ring-light types, each at most a cooperative part and a remainder
    int n;
    unsigned coop_mask;
    int count[kMax];
    int offset[kMax];
    __host__ __device__ int blocks() const
    {
        int b = 0;
        for (int k = 0; k < n; ++k) b += (count[k] + 3) >> 2;
        return b;
    }
};

// Flags of one launch: which duties of the mix loop this slot's kernel performs.
enum : int {
    kFirst = 1, // slot 0: start from the dry mix of the input instead of reading mixbuf
    kLast = 2,  // last slot: write the interleaved output instead of mixbuf
    kNoCuMajor = 32, // experiment (OALSFX_DEBUG_FLAGS 0x100): grids of several kinds in plain workgroup order
    kFilterInside = 64, // the steady-state reverb builds that have the send filters inside (SF) apply them themselves for the instances that
                        // have one switched on: the pre-pass leaves those instances out
    kFiltered = 16, // the send-filter pre-pass (k_send_filters) ran: for the instances with a filter switched on, the planes at
                    // filtered_src / wet planes hold their sends' inputs and their filter histories are up to date
};

constexpr int kWave = 64;
// The plain build of the steady-state reverb kernel takes instances whose every tap is at least this many samples from where it is
// written (late taps: from the late feed): three tiles, so that the aligned windows it requests a tile ahead (reverb.hip, OALSFX_AW)
// never reach samples that are still being written.  The host sorts proven instances into kinds by the same number (batch.cpp).
constexpr unsigned kPlainMinTap = 192;
// ... and its early taps and late-line offsets 32 samples more: those two groups' windows are requested before the tile's stores to
// their rings are issued, and the line-aligned stores (reverb.hip, CR) hold a tile's last samples back by up to 31
constexpr unsigned kPlainMinTapAhead = 224;

// ---- launchers (defined next to their kernels) ----
// The host splits every reverb list into the instances it believes steady and the rest (a speed hint: the steady-state
// kernel still decides per instance from the device state).  close_taps: some listed instance has a tap distance of 64..127
// samples, which selects the kernel build that can request such groups late; modulated: some listed instance has (or had)
// a modulated late line, which selects the build that carries the modulation; short_taps: some listed instance has a tap
// shorter than one tile, which selects the most general build.
// proven: the host has device-confirmed knowledge that every listed instance is steady (mono / stereo, whole tiles): the FP builds.
// in_transition: some listed instance had its properties changed less than a cross-fade ago, in a way the XF build can follow (mono /
// stereo, whole tiles, not proven): that build.
// Returns the kernel symbol it launched (template arguments as rocprofv3 prints them), nullptr when the list was empty.
// carry: the write position of some listed instance is off the 128-byte line grid of its delay lines (an odd-sized call came before):
// the plain FP build that writes every ring line in whole cache lines (reverb.hip, CR == 2).
// groups (may be nullptr): receives the number of workgroups launched -- each counts itself in at ctx.turn_started, and the host's gate
// in front of the next chained launch is set by that number (batch.cpp, started_total).
const char* launch_reverb_steady(const KernelCtx& ctx, int slot, const int* list, int count, int flags, bool close_taps, bool modulated,
                                 bool short_taps, bool proven, bool in_transition, hipStream_t stream, int* groups = nullptr, bool carry = false);
// The steady reverbs of a slot listed by kind (mono / stereo, whole tiles): counts[0] proven, every tap two tiles away; [1] proven, a tap
// of one to two tiles; [2] proven, shorter taps or a modulated late line; [3] believed steady or in a transition the XF build follows.
// One kind alone runs its own lean kernel, several share one grid whose workgroups take the build of their kind.
// no_fallback: the believed kind without the general path inside (the host predicts the kernel's test; a miss is counted in ctx.fault).
// filters_inside: some instance of the first two kinds has a send filter switched on and the batch has one slot: those kinds run the SF
// builds (send filters inside, flag kFilterInside), and the pre-pass need not know them.
const char* launch_reverb_steady_kinds(const KernelCtx& ctx, int slot, const int* list, const int counts[4], int flags, bool no_fallback, bool filters_inside,
                                       hipStream_t stream, int* groups = nullptr, bool carry = false);
void launch_reverb_general(const KernelCtx& ctx, int slot, const int* list, int count, int flags, hipStream_t stream);
// every ring-light effect type of `slot_count` consecutive slots in one grid, one wavefront per listed instance (wave_effects.hip)
// `seg` (single slots only, may be nullptr): the grid follows the list segment by segment, see WaveSegments
void launch_wave_effects(const KernelCtx& ctx, int slot, int slot_count, const int* list, int count, const WaveSegments* seg, int flags,
                         hipStream_t stream);
// mono / stereo, whole tiles: the believed-steady reverbs and the ring-light effects of one slot in one grid (reverb.hip)
// (`proven`: every listed reverb is proven steady and the call's blocks leave their gains at rest: the FP build of the reverb groups)
void launch_slot_mixed(const KernelCtx& ctx, int slot, const int* steady_list, int steady_count, const int* light_list, int light_count,
                       const WaveSegments& seg, int flags, bool proven, hipStream_t stream, int* groups = nullptr);
// Send shelf filters of every instance (reference apply_filters, src/oalsfxpp.cpp:3101-3143): reads `src`, writes the direct
// send's input to filtered[0] and slot s's to filtered[1 + s], each [instance][frames][channels] with stride ctx.src_stride.
// `list` (may be nullptr: instances 0 .. instances - 1): the instances to look at (those among them without a filter are skipped)
// The fault word (KernelCtx::fault, host memory): instances a proven-steady launch had to leave alone count 1 each; waits of chained
// launches that gave up count in fields of their own
constexpr unsigned kFaultTurn = 1u << 12; // an instance's turn did not come (reverb.hip)
constexpr unsigned kFaultGate = 1u << 24; // k_chain_gate counted out

// One wavefront that waits until `*started` has reached `target` (the chained launch before has all but a few of its workgroups on the
// chip): queued in front of every chained launch but a run's first.
void launch_chain_gate(const unsigned* started, unsigned target, unsigned* fault, hipStream_t stream);
void launch_send_filters(const KernelCtx& ctx, const float* src, long long src_stride, float* filtered, size_t send_floats, const int* list, int instances,
                         hipStream_t stream);
// record k of `packed` (count records of record_bytes, a multiple of 4) goes to slot indices[k] of the device array `dst`
void launch_scatter_records(void* dst, size_t record_bytes, const void* packed, const int* indices, int count, hipStream_t stream);
// A whole parameter upload in one launch: record k of `packed` goes to slot indices[k] of `dst` (four arrays), and two plain copies
// (dword counts; 16-byte aligned).  Sources may be page-locked host memory.
struct ScatterJob { unsigned* dst; const unsigned* packed; const int* indices; int record_dwords; int count; int takes_turns; };
struct CopyJob { unsigned* dst; const unsigned* src; size_t dwords; int blocks; };
// turn / turn_wait / fault (chained launches, batch.cpp): a record of a job that takes turns is stored once the launch before is through
// with the instance it belongs to (the word turn[indices[k]] holds turn_wait); nullptr / 0: at once
// gate_started / gate_target: the upload doubles as the gate of the chained launch behind it (k_chain_gate): its first workgroup also
// waits until *gate_started has reached gate_target
struct UploadJobs { ScatterJob scatter[4]; CopyJob copy[2]; const unsigned* turn; unsigned turn_wait; unsigned* fault; const unsigned* gate_started; unsigned gate_target; };
void launch_upload(UploadJobs jobs, hipStream_t stream);
void launch_null(hipStream_t stream);
// dst[0 .. floats) = src[0 .. floats), either of them possibly page-locked host memory mapped into the device's address space
void launch_copy_floats(float* dst, const float* src, size_t floats, hipStream_t stream);
void launch_fill_synthetic(float* dst, int instances, int floats_per_instance, unsigned buffer_index, hipStream_t stream);
void launch_ring_probe(float* slabs, int instances, size_t slab_floats, unsigned pos0, int waves_per_slab, hipStream_t stream);
void launch_stream_pattern(float* slabs, int instances, int dwords_per_lane, unsigned pos0, size_t slab_floats, int pos_skew, hipStream_t stream);
void launch_hbm_sweep(float* buf, size_t floats, int write, float* sink, hipStream_t stream);

// Launches of this host thread from here on ask for as much LDS per workgroup as this many bytes (0: what the kernel declares).  A step of
// two different kernels whose launches overlap (batch.cpp: chain_eligible) needs workgroups of one size: a CU hands out LDS in contiguous
// blocks (and registers likewise: equal_places below), and a workgroup of one kernel does not fit the hole a smaller one of the other
// left -- measured: the ring-light kernel with three of its four workgroups per CU on the chip, 113 us per step instead of 94.
void set_lds_per_workgroup(int bytes);
int lds_per_workgroup();

#if defined(__HIPCC__)

template <class K> inline int declared_lds(K kernel)
{
    hipFuncAttributes a{};
    return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(kernel)) == hipSuccess ? static_cast<int>(a.sharedSizeBytes) : 0;
}

#define OALSFX_LAUNCH(kernel, grid, block, stream, ...)                                                          \
    do {                                                                                                         \
        int more_lds_ = 0;                                                                                       \
        if (oalsfx_hip::lds_per_workgroup() > 0) {                                                               \
            static const int declared_ = oalsfx_hip::declared_lds(kernel);                                       \
            more_lds_ = (declared_ > 0 && oalsfx_hip::lds_per_workgroup() > declared_) ? oalsfx_hip::lds_per_workgroup() - declared_ : 0; /* (0: the runtime did not say) */ \
        }                                                                                                        \
        hipLaunchKernelGGL(kernel, grid, block, more_lds_, stream, __VA_ARGS__);                                 \
    } while (0)

// A kernel whose wavefronts must take up exactly 128 registers, whatever the build needs (see set_lds_per_workgroup; used by the variants
// of the reverb builds that a two-kernel step launches, k_reverb_steady_coop_ep, and by the mono ring-light kernel): the last statement
// of the kernel, where nothing is live (as the first it cost several builds a spill).
#define OALSFX_EQUAL_PLACES() asm volatile("" ::: "v127")

// A grid whose workgroups run different code (k_reverb_steady_kinds: one build per kind of instance; the ring-light grids: one body per
// effect type) is ordered by kind, and consecutive workgroups land on consecutive CUs: workgroup g of a grid of 256-thread workgroups
// that are all resident at once goes to CU slot g % 256 (measured, scripts/micro/placement.hip), so every CU would run every kind and
// its instruction cache would hold none of them (measured: 4096 reverbs of four kinds 73 us per buffer against 57 with one kind).
// This maps g to its position in CU-major order -- the workgroups that share a CU first, then the next CU -- so that a kind's
// workgroups fill whole CUs.  Rounds of 1024 workgroups (four per CU); a bijection on [0, total) for any total.
__device__ __forceinline__ int cu_major_position(int g, int total)
{
    const int base = g & ~1023;
    const int n = min(total - base, 1024), l = g - base;
    const int rounds = (n + 255) >> 8, rem = n - ((rounds - 1) << 8); // rem: how many CU slots have a workgroup in the last round
    int c = l & 255;
    const int r = l >> 8;
#ifdef OALSFX_CU_ORDER_SE
    // experiment: CU slots in shader-engine-major order (XCC, SE, CU): c = xcc + 8 * (se + 4 * cu)  ->  rank ((xcc * 4 + se) * 8 + cu)
    if (n == 1024) c = (((c & 7) * 4 + ((c >> 3) & 3)) << 3) | (c >> 5);
#endif
    return base + (c < rem ? c * rounds + r : c * (rounds - 1) + rem + r);
}

__device__ __forceinline__ bool audible(float g) { return fabsf(g) > OALSFX_SILENCE_GAIN; }

// The CU this wavefront runs on: XCC_ID, and shader engine / array / CU of HW_ID; never 0.
__device__ __forceinline__ unsigned this_cu()
{
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    return 0x10000u | ((xcc & 15u) << 8) | ((hw >> 8) & 0xFFu);
}

// Chained launches for a kernel with one wavefront per instance (wave_effects.hip); the rules, and why each step is there, are written
// down where the steady-state reverb kernel does the same by hand (reverb.hip, "Chained launches").  turn_take: waits until the
// launch before is through with the instance whose word is turn[word]; false when the turn never came (counted in the fault word: the
// wavefront must then leave the instance alone, word included).  Nothing of the instance may be read before it returns.
// dbg: the test switches 1 (always pay for the acquire) and 2 (never).
__device__ __forceinline__ bool turn_take(const KernelCtx& ctx, size_t word, int lane, int dbg, unsigned& cu_before)
{
    cu_before = 0;
    if (ctx.turn == nullptr || ctx.turn_wait == 0u) return true;
    int lost = 0;
    if (lane == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(ctx.turn + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ctx.turn_wait) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 20)) {
                if (ctx.fault) __hip_atomic_fetch_add(ctx.fault, kFaultTurn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                lost = 1;
                break;
            }
        }
    }
    if (__builtin_amdgcn_readfirstlane(lost)) return false;
    unsigned before_cu = 0, before_that_cu = 0;
    if (lane == 0) {
        before_cu = __hip_atomic_load(ctx.turn_cu + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        before_that_cu = __hip_atomic_load(ctx.turn_cu2 + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    before_cu = __builtin_amdgcn_readfirstlane(before_cu);
    before_that_cu = __builtin_amdgcn_readfirstlane(before_that_cu);
    cu_before = before_cu;
    if (!(dbg & 2) && ((dbg & 1) || before_cu == this_cu() || (ctx.turn_two_back != 0u && before_that_cu == this_cu()))) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (ctx.turn_started != nullptr && lane == 0) __hip_atomic_fetch_add(ctx.turn_started + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (a count for the records)
    } else {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // (no instruction: keeps the loads behind the wait)
    }
    __builtin_amdgcn_s_dcache_inv();
    __builtin_amdgcn_s_waitcnt(0);
    return true;
}

// ... and hands the instance on: every store of the wavefront acknowledged, then the CU names, then the word.
__device__ __forceinline__ void turn_hand_on(const KernelCtx& ctx, size_t word, int lane, unsigned cu_before)
{
    if (ctx.turn == nullptr || ctx.turn_set == 0u) return;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) {
        __hip_atomic_store(ctx.turn_cu2 + word, cu_before, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ctx.turn_cu + word, this_cu(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_s_waitcnt(0); // vmcnt(0) expcnt(0) lgkmcnt(0)
    if (lane == 0) __hip_atomic_store(ctx.turn + word, ctx.turn_set, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ float lerpf(float a, float b, float mu) { return a + ((b - a) * mu); }

// ---------------------------------------------------------------------------------------------
// sinf with the results of the reference's libm (glibc 2.35 x86-64 FMA build).  The reference
// truncates sin()-derived values to integer delay taps inside process loops (reference
// src/oalsfxpp.cpp:4273, 7454-7465), so the device must reproduce that libm bit for bit rather than
// use the device math library.  Algorithm: quadrant reduction and polynomial in double precision
// (glibc sysdeps/ieee754/flt-32/s_sinf.c, ARM optimized-routines), every a*b+c fused.
// Valid for |y| < 120, which covers every argument the process path forms.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float glibc_sinf_poly(double x, double x2, bool negate_cos, int n)
{
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1 = fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
        const double x7 = x3 * x2;
        const double s = fma(x3, -0x1.555545995a603p-3, x);
        return (float)fma(x7, s1, s);
    }
    const double sg = negate_cos ? -1.0 : 1.0;
    const double x4 = x2 * x2;
    const double c2 = fma(x2, sg * 0x1.99343027bf8c3p-16, sg * -0x1.6c087e89a359dp-10);
    const double c1 = fma(x2, sg * -0x1.ffffffd0c621cp-2, sg * 0x1p0);
    const double x6 = x4 * x2;
    const double c = fma(x4, sg * 0x1.55553e1068f19p-5, c1);
    return (float)fma(x6, c2, c);
}

__device__ __forceinline__ float glibc_sinf(float y)
{
    const uint32_t top = (__float_as_uint(y) >> 20) & 0x7ff;
    double x = (double)y;
    if (top < 0x3f4) {            // |y| < 0.75 region the libm treats without reduction (abstop12(pi/4))
        if (top < 0x398) return y; // |y| < 2^-12
        return glibc_sinf_poly(x, x * x, false, 0);
    }
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int32_t)r + 0x800000) >> 24;
    x = fma(-(double)n, 0x1.921FB54442D18p0, x);
    const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
    return glibc_sinf_poly(x * s, x * x, (n & 2) != 0, n);
}

// lround() for floats: nearest, halfway cases away from zero.
__device__ __forceinline__ int lround_away(float x)
{
    float t = truncf(x);
    if (fabsf(x - t) >= 0.5f) t += copysignf(1.0f, x);
    return (int)t;
}

// Direct-form-I biquad step with the reference's association (reference FilterState::process,
// src/oalsfxpp.cpp:1009-1014).
__device__ __forceinline__ float biquad_step(const oalsfx_biquad_t& c, oalsfx_hist_t& h, float x)
{
    const float y = (c.b0 * x) + (c.b1 * h.x[0]) + (c.b2 * h.x[1]) - (c.a1 * h.y[0]) - (c.a2 * h.y[1]);
    h.x[1] = h.x[0];
    h.x[0] = x;
    h.y[1] = h.y[0];
    h.y[0] = y;
    return y;
}

// Does any enabled send of this instance have a shelf filter switched on?  The send-filter pre-pass handles exactly those
// instances (all their sends); the effect kernels read the pre-pass planes for them and the raw input for the others.
__device__ __forceinline__ bool instance_has_send_filter(const KernelCtx& ctx, int inst)
{
    typedef const __attribute__((address_space(4))) oalsfx_source_params ConstSourceParams;
    ConstSourceParams& P = *(ConstSourceParams*)(uintptr_t)(ctx.source + inst);
    int any = P.direct.filter_type;
    for (int s = 0; s < ctx.slots; ++s)
        if (P.aux[s].out_channels != 0) any |= P.aux[s].filter_type;
    return any != OALSFX_AF_NONE;
}

// Pass-through sends (ActiveFilters::none): the histories of both shelf filters of every enabled send follow the input
// (reference process_pass_through, src/oalsfxpp.cpp:1038-1056; disabled sends of null slots are skipped, :2952-2956).
// Called for input channel c by the kernel that owns the kFirst duty when no filter pre-pass ran.
__device__ __forceinline__ void send_history_follow(const KernelCtx& ctx, int inst, int c, int channels, int frames, const float* src)
{
    if (frames <= 0) return;
    const oalsfx_source_params& P = ctx.source[inst];
    oalsfx_source_state& S = ctx.source_state[inst];
    const float newest = src[static_cast<size_t>(frames - 1) * channels + c];
    const float older = frames >= 2 ? src[static_cast<size_t>(frames - 2) * channels + c] : 0.0F;
    for (int send = 0; send <= ctx.slots; ++send) {
        if (send > 0 && P.aux[send - 1].out_channels == 0) continue;
        oalsfx_hist_t* h[2] = {&S.lp[send][c], &S.hp[send][c]};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (frames >= 2) {
                h[k]->x[1] = older; h[k]->y[1] = older;
            } else {
                h[k]->x[1] = h[k]->x[0]; h[k]->y[1] = h[k]->y[0];
            }
            h[k]->x[0] = newest; h[k]->y[0] = newest;
        }
    }
}

// The same for a kernel that already knows which sends are enabled (bit 0 the direct send, bit 1 + s the send to slot s) and holds
// the call's last two frames of channel c in registers (calls of two frames or more): no loads at the end of the launch.
__device__ __forceinline__ void send_history_follow_values(const KernelCtx& ctx, int inst, int c, unsigned send_mask, float newest, float older)
{
    oalsfx_source_state& S = ctx.source_state[inst];
    for (int send = 0; send <= ctx.slots; ++send) {
        if (!((send_mask >> send) & 1u)) continue;
        oalsfx_hist_t* h[2] = {&S.lp[send][c], &S.hp[send][c]};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            h[k]->x[1] = older; h[k]->y[1] = older;
            h[k]->x[0] = newest; h[k]->y[0] = newest;
        }
    }
}

// Synthetic benchmark input, identical to oracle_synth.
__device__ __forceinline__ uint32_t synth_seed(uint32_t instance, uint32_t buffer_index)
{
    uint32_t x = 0x9E3779B9u ^ (instance * 2654435761u) ^ buffer_index;
    return x == 0 ? 1u : x;
}

#endif // __HIPCC__

} // namespace oalsfx_hip

#endif
