// Reverb / EAX reverb process kernels for gfx950 (the headline path).
//
// They replace ReverbEffectState::do_process and everything under it (reference src/oalsfxpp.cpp:6078-6170, 7358-7903)
// plus, when the slot is first / last, the dry mix of mix_source (src/oalsfxpp.cpp:2917-2950) and the interleave of
// write_f32 (src/oalsfxpp.cpp:3414-3431).
//
// Map of this file (design notes in DESIGN.md 3.1):
//   * helpers shared by both kernels: byte-offset ring addressing, the scattering matrix, the serial halves of the
//     biquad and first-order sections (one LDS request ahead of the dependent arithmetic);
//   * k_reverb_steady_coop -- instances in their steady state: a workgroup is four wavefronts = four instances, lanes are
//     64 consecutive sample times, all ring taps of a tile are requested one tile ahead, and the serial filter
//     recurrences of the four instances run together on 16 lanes of one wavefront per chain phase.  Builds: plain,
//     HY (taps of 64..127 samples), MD (modulated late line), ST (taps shorter than a tile);
//   * reverb_general_instance / k_reverb -- everything else (cross-fades, gain ramps, ragged tiles, more than two
//     channels, parameter sets outside the steady-state builds): one wavefront per instance, feedback honoured by
//     cutting a tile into sub-blocks no longer than the shortest positive feedback delay; also the out-of-line fallback
//     of the steady-state kernel for an instance that turns out not to be steady;
//   * the launchers.
//
// Common to both: every delay ring is stored per line (line j of ring r is one contiguous power-of-two ring), so a tap
// read or a ring write of a tile is one contiguous 256-byte wave access; each lane carries the 4-line vector of its
// sample in registers; recurrences that must round exactly like the reference keep their order (feed-forward half per
// lane, feedback half on chain lanes over an LDS transpose).
//
// Ordering of ring stores and later tap loads inside one wavefront relies on the hardware executing a wavefront's
// memory instructions in order; the wavefront-scope fences only pin the compiler.
//
// Bit-exactness: compiled with -ffp-contract=off; expression association follows the reference.
#include <float.h>

#include "common.hpp"
#include "wave_effects_body.hpp"

namespace oalsfx_hip {

namespace {

constexpr int kRow = 68;          // 4 (history prefix, 16-byte aligned data) + 64 samples
constexpr int kGroups = 3;        // row groups, 4 lines each (general path)
constexpr int kSteadyGroups = 6;  // the steady-state kernel: three for the input half of a tile (shelves), three for its late half (T60)
constexpr int kRngFloats = OALSFX_RV_MAX_UPDATE;

// Per-wave table of instance constants kept in LDS (dword offsets).  The steady-state tile reads them
// with broadcast ds_reads right where they are used: they cost no VALU slot and no long-lived SGPRs.
namespace ut {
enum {
    TAP4 = 0,    // 6 groups x 4 lines: current taps as byte distances (early tap, early AP, early line, late tap, late AP, late line)
    LO = 24,     // 5 rings x 4 lines: byte offset of each line in the slab
    BMASK = 44,  // 5 rings: (len - 1) * 4
    FEED4 = 49,  // late_feed_tap * 4
    CR_R = 50, CR_RF = 51, // line-aligned stores (CR builds): samples the call's write position / its late feed position lie past a 128-byte line
    ECOEF = 52, ELCOEF = 56,
    MISC = 60,   // density_gain, ap_feed_coeff, mix_x, mix_y
    LPB = 64, HPB = 68,        // b0, b1, b2 of the input shelves
    TL0 = 72, TL1 = 76, TH0 = 80, TH1 = 84, // T60 feed-forward coefficients per line
    GOUT = 88,   // 16 output gains of the chunk, chain order q = (stage*4 + line)*2 + channel (stereo / mono only)
    GDIR = 104,  // direct gains [in][out] (2x2)
    GAUX = 108,  // aux gains [in][k] (2x4)
    TL2 = 116, TH2 = 120, TMID = 124, // T60 feedback coefficients and mid gains per line (chain lanes)
    SIZE = 128
};
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int CH>
struct Lds {
    static constexpr int kChains = 8 * CH; // (early|late) x 4 lines x channels gain ramps
    static constexpr int kFloats = kGroups * 4 * kRow + kRngFloats + kChains * 64 + ut::SIZE;
};

// packed form of vector_partial_scatter on (v0,v1 | v2,v3); same association as the scalar form
__device__ __forceinline__ void scatter2(v2f& a, v2f& b, float x, float y)
{
    const float v0 = a.x, v1 = a.y, v2 = b.x, v3 = b.y;
    v2f s0 = v2f{v1, -v0} + v2f{-v2, v2};
    s0 = s0 + v2f{v3, v3};
    v2f s1 = v2f{v0, -v0} + v2f{-v1, -v1};
    s1 = s1 + v2f{v3, -v2};
    a = (x * a) + (y * s0);
    b = (x * b) + (y * s1);
}

// Hand-off between lanes of one wavefront through LDS or through the rings in global memory:
// pins the compiler; the hardware keeps one wavefront's memory operations in program order.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int min_positive(int limit, int d) { return (d > 0 && d < limit) ? d : limit; }
__device__ __forceinline__ int min4(const int v[4]) { return min(min(v[0], v[1]), min(v[2], v[3])); }

// A ring in byte terms: line j of the ring starts `lo[j]` bytes into the slab and is `bmask + 4` bytes long.
// Rings are placed longest-first (oalsfx_reverb_place_rings), so every line starts at a multiple of its
// own length and "line base | wrapped position" is one AND-OR on a 32-bit byte offset.
struct Ring {
    unsigned lo[4];
    unsigned bmask;
    __device__ __forceinline__ unsigned at(int j, unsigned pos4) const { return (pos4 & bmask) | lo[j]; }
};

// The slab pointer comes out of a table in memory, so the compiler cannot infer its address space; saying
// "global" explicitly turns every ring access into global_load/store with a scalar base and a 32-bit offset.
typedef __attribute__((address_space(1))) char GlobalBytes;
__device__ __forceinline__ float ld(const GlobalBytes* slab, unsigned byte_off)
{
    return *reinterpret_cast<const __attribute__((address_space(1))) float*>(slab + byte_off);
}
__device__ __forceinline__ void st(GlobalBytes* slab, unsigned byte_off, float v)
{
    *reinterpret_cast<__attribute__((address_space(1))) float*>(slab + byte_off) = v;
}
// Non-temporal hints on the ring traffic of the steady-state kernel.  Bits of OALSFX_NT: 1 stores to the three long rings (main delay,
// early line, late line: two thirds of the traffic, re-read tens of launches later at the earliest), 2 loads from them, 4 stores to the
// two all-pass rings (96 MiB for 4096 instances, re-read within three launches), 8 loads from those, 16 the output frames, 32 the input
// frames.  Measured in round 3 on the headline workload, each against the build without hints in one process (scripts/ab_nt.sh,
// profiles/r03c_nontemporal/): 1: -4.1 %, 2: +-0, 1|2: -6.3 ... -6.8 % (47.1 -> 44.2 us), 1|4: -5.1 %, 1|2|4: +1.0 %, all four: +9.5 %,
// 1|2|16: +3 % against 1|2, 1|2|32: -0.6 %.  What streams through for good is told so and leaves the caches to the all-pass rings,
// which come back within a launch or three; hinting those too sends them to memory and back.  Default: 1|2.
// (Also tried: the lanes of a wavefront's last, only begun 128-byte line without the hint, so that the next tile finds it cached: +2.1 %.
// With the hints the cost of the unaligned taps -- every interior line fetched twice -- shows more: taps rounded to lines, an
// ablation with wrong results, now gain 11 % where they gained 5 %.)
#ifndef OALSFX_NT
#define OALSFX_NT 3
#endif
// Aligned windows for the long rings' taps (the plain FP build: every tap at least two tiles away, requested a tile ahead).  A tap's 64
// samples start anywhere in a 128-byte line, so a wavefront's request touches three lines and every interior line is fetched by two
// consecutive tiles; with the non-temporal hints nothing keeps it in a cache in between.  Instead: request the 256-byte-aligned window
// that holds the end of what the tile needs, keep the window before it in a register per stream, and take each lane's sample from one
// of the two with a lane rotation (ds_bpermute).  Two lines per stream and tile instead of three, one window more per launch.
// The window requested for a tile reaches up to 63 samples past what the tile needs -- samples the *next* tile needs --, and it is
// requested a tile ahead: so every tap of such an instance must be three tiles away (kPlainMinTap, which host and device share).
// Measured (scripts/ab_libs.py, profiles/r03g_aligned_windows/): 256-frame calls 43.9 -> 42.3 us (-3.5 %), 512-frame calls -8.5 %,
// 2048-frame calls 333.8 -> 297.6 us (-10.9 %: 37.2 us per 256 frames, 0.73 of the roofline).
#ifndef OALSFX_ABLATE_VALU
#define OALSFX_ABLATE_VALU 0
#endif
#ifndef OALSFX_CHAIN_EXP
#define OALSFX_CHAIN_EXP 0 // experiments: 1 an agent-scope acquire behind every wait for a turn, 2 plain stores of the output frames (timing only),
                           // 4 the word as an agent-scope release store (an L2 write-back in front of it),
                           // 8 no wait at all (negative control of tests/test_gpu_chained.py: it must fail)
#endif
#ifndef OALSFX_AW
#define OALSFX_AW 1
#endif
#ifndef OALSFX_CU_CHECK_BESIDE_RECORD
#define OALSFX_CU_CHECK_BESIDE_RECORD 1 // chained launches, FP builds: the CU names of the launches before travel beside the hot record (0: in front of it, as before)
#endif
#ifndef OALSFX_EARLY_HANDBACK
#define OALSFX_EARLY_HANDBACK 1 // FP builds write state and hot record in front of the last tile's S5 instead of behind the loop (0: as before, same-box A/B)
#endif
#ifndef OALSFX_CR_FEED
#define OALSFX_CR_FEED 0 // 1: the plain FP builds for write positions on the grid hold the late feed's stores back too (CR == 1; measured: no gain)
#endif
static_assert(!OALSFX_AW || kPlainMinTap >= 192, "aligned windows: a window requested a tile ahead may reach 63 samples past its tile's taps");
template <int R> struct RingId { static constexpr int value = R; };
constexpr bool long_ring(int r) { return r == OALSFX_RV_MAIN || r == OALSFX_RV_EARLY_LINE || r == OALSFX_RV_LATE_LINE; }
// (the ring is a template argument: decided at run time, the two loads of one address are merged before the ring is known and the hint is lost)
template <int R>
__device__ __forceinline__ float ld_ring(const GlobalBytes* slab, unsigned byte_off)
{
    if constexpr (((OALSFX_NT & 2) && long_ring(R)) || ((OALSFX_NT & 8) && !long_ring(R)))
        return __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) float*>(slab + byte_off));
    else
        return ld(slab, byte_off);
}
template <int R>
__device__ __forceinline__ void st_ring(GlobalBytes* slab, unsigned byte_off, float v)
{
    if constexpr (((OALSFX_NT & 1) && long_ring(R)) || ((OALSFX_NT & 4) && !long_ring(R)))
        __builtin_nontemporal_store(v, reinterpret_cast<__attribute__((address_space(1))) float*>(slab + byte_off));
    else
        st(slab, byte_off, v);
}

// delay_out_faded / delay_out_unfaded (reference src/oalsfxpp.cpp:7358-7399); pos4 = 4 * sample position
__device__ __forceinline__ float tap(const GlobalBytes* slab, const Ring& r, int j, bool faded, unsigned pos4_0, unsigned pos4_1, float mu)
{
    const float a = ld(slab, r.at(j, pos4_0));
    if (!faded) return a;
    const float b = ld(slab, r.at(j, pos4_1));
    return lerpf(a, b, mu);
}

// vector_partial_scatter (reference src/oalsfxpp.cpp:7510-7521)
__device__ __forceinline__ void scatter(float v[4], float x, float y)
{
    const float f0 = v[0], f1 = v[1], f2 = v[2], f3 = v[3];
    v[0] = (x * f0) + (y * (f1 + -f2 + f3));
    v[1] = (x * f1) + (y * (-f0 + f2 + f3));
    v[2] = (x * f2) + (y * (f0 + -f1 + f3));
    v[3] = (x * f3) + (y * (-f0 + -f1 + -f2));
}

// Output gains at rest.  The reference ramps a gain over the first frames of a block when |target - current| / frames of the block
// exceeds FLT_EPSILON and leaves it alone otherwise (for good, if the target is that close: src/oalsfxpp.cpp:2752-2798), so a current
// gain may sit a few millionths off its target forever -- until a call with a shorter block comes along.  The blocks of a call of
// whole tiles are 64, 128, 192 or 256 frames long.  Returns the shortest of those lengths whose blocks (and all longer ones) leave
// every gain of the instance alone, 0 if even a 256-frame block would ramp one (a vote of the whole wavefront: `counts` says which
// lanes hold a gain).  Same expression as the kernels' own "is a ramp in flight" tests.
__device__ __forceinline__ unsigned gains_rest_level(bool counts, float current, float target)
{
    const float d = target - current;
    if (__ballot(counts && fabsf(d * (1.0F / 64.0F)) > FLT_EPSILON) == 0ULL) return 64u;
    if (__ballot(counts && fabsf(d * (1.0F / 128.0F)) > FLT_EPSILON) == 0ULL) return 128u;
    if (__ballot(counts && fabsf(d * (1.0F / 192.0F)) > FLT_EPSILON) == 0ULL) return 192u;
    if (__ballot(counts && fabsf(d * (1.0F / 256.0F)) > FLT_EPSILON) == 0ULL) return 256u;
    return 0u;
}

// Serial half of a biquad over samples [0, n) of one LDS row: y = (u - a1*y1) - a2*y2.
// row_u[4+i] holds the feed-forward sums, row_y[4+i] receives the outputs.
__device__ __forceinline__ void biquad_chain(const float* row_u, float* row_y, int n, float a1, float a2, float& y1, float& y2)
{
    int i = 0;
    if (n >= 4) {
        // the next four sums are requested before the current four are worked on: the LDS latency hides behind the
        // dependent arithmetic instead of adding to it (rows may alias: the request is ahead of the store)
        float4 u = *reinterpret_cast<const float4*>(row_u + 4);
        for (; i + 4 <= n; i += 4) {
            float4 un = u;
            if (i + 8 <= n) un = *reinterpret_cast<const float4*>(row_u + 8 + i);
            float4 y;
            y.x = (u.x - (a1 * y1)) - (a2 * y2);
            y.y = (u.y - (a1 * y.x)) - (a2 * y1);
            y.z = (u.z - (a1 * y.y)) - (a2 * y.x);
            y.w = (u.w - (a1 * y.z)) - (a2 * y.y);
            *reinterpret_cast<float4*>(row_y + 4 + i) = y;
            y2 = y.z;
            y1 = y.w;
            u = un;
        }
    }
    for (; i < n; ++i) {
        const float y = (row_u[4 + i] - (a1 * y1)) - (a2 * y2);
        row_y[4 + i] = y;
        y2 = y1;
        y1 = y;
    }
}

// Serial half of a first-order section over samples [lo, hi): o = u + c2*o_prev; stores scale*o.
__device__ __forceinline__ void first_order_chain(const float* row_u, float* row_o, int lo, int hi, float c2, float scale, bool scaled, float& prev)
{
    int i = lo;
    if ((lo & 3) == 0 && lo + 4 <= hi) {
        float4 u = *reinterpret_cast<const float4*>(row_u + 4 + lo); // one request ahead, as in biquad_chain
        for (; i + 4 <= hi; i += 4) {
            float4 un = u;
            if (i + 8 <= hi) un = *reinterpret_cast<const float4*>(row_u + 8 + i);
            float4 o;
            o.x = u.x + (c2 * prev);
            o.y = u.y + (c2 * o.x);
            o.z = u.z + (c2 * o.y);
            o.w = u.w + (c2 * o.z);
            prev = o.w;
            if (scaled) { o.x = scale * o.x; o.y = scale * o.y; o.z = scale * o.z; o.w = scale * o.w; }
            *reinterpret_cast<float4*>(row_o + 4 + i) = o;
            u = un;
        }
    }
    for (; i < hi; ++i) {
        const float o = row_u[4 + i] + (c2 * prev);
        prev = o;
        row_o[4 + i] = scaled ? scale * o : o;
    }
}

} // namespace


// =================================================================================================
// Cooperative steady-state kernel: one wavefront per instance, 64 sample times per tile, packed arithmetic, taps requested
// a tile ahead; the serial filter recurrences of the four instances of a workgroup are run together: for every chain
// phase one wavefront executes the recurrences of all 16 (instance, line) pairs on 16 lanes while its siblings wait at a
// workgroup barrier.  Run per wavefront on 4 lanes those recurrences were about 70 % of the vector instructions; sharing
// them cuts that part by four.  An instance that is not in its steady state sits the phases out (empty chain range) and
// takes the general path at the end of the kernel.
//
// Barriers are raw s_barrier preceded by s_waitcnt lgkmcnt(0) only: LDS traffic must be complete, global
// loads (the prefetch of the next tile) and ring stores stay in flight across them.
// =================================================================================================
template <int CH>
__device__ __noinline__ void reverb_general_call(const KernelCtx* ctx, int slot, int inst, int flags, float* lds, int lane);

namespace coop {
enum { LPX0, LPX1, LPY0, LPY1, HPY0, HPY1, T60X, T60O1, T60O2, LP_A1, LP_A2, HP_A1, HP_A2, T_L2, T_H2, T_MID, SIZE };
}

__device__ __forceinline__ int wib_of(unsigned tid) { return __builtin_amdgcn_readfirstlane(static_cast<int>(tid >> 6)); }

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// HY ("hybrid" requests): taps between one and two tiles are accepted too; their groups are requested at the top of
// their own tile instead of one tile ahead.  Chosen per launch by the host from the parameters (a speed hint).
// MD (implies HY): instances with a modulated late line are accepted: the depth smoother (a serial lerp chain) runs on
// lane 0 of each wavefront one tile ahead, the per-sample delays shift the late-line requests.
// ST (implies HY and MD): taps shorter than a tile where the data they read is produced inside the tile without a loop
// through the reverb core, or where a few extra evaluations settle it: early taps, late taps and early-line offsets of any
// length (their sources -- the filtered input, the late feed, the reversed all-pass output of an earlier lane -- are handed
// over through LDS rows), and vector all-pass offsets of 4..63 samples (the all-pass outputs of the first lanes are
// evaluated ahead, 63 / offset times, and handed over; offsets under 16 only occur below 16 kHz).
// CH == 8: the multichannel build (quad .. 7.1, channel count at run time): send and pan gains live in a second table,
// the dry mix and the panning loop over the channels, and an instance that is not steady is not taken inside the kernel
// (its LDS would have to be sized for the general path's 64 gain ramps) but left, through ctx.progress, to the general
// kernel that the host launches right after on the same list (any build works that way when ctx.progress is set).
// XF (a variant of the ST build, mono / stereo, whole tiles, not FP): an instance whose properties were changed stays in the cooperative
// workgroup while it cross-fades its taps and ramps its output gains (reference src/oalsfxpp.cpp:6062-6075, 6088-6096, 6118-6138,
// 7378-7399, 2752-2798) -- provided both tap sets are ones the ST build accepts and the fade stands at a tile boundary.  Its
// cross-fading tiles request the taps being faded in at the top of the tile, behind the previous tile's stores, and mix the two sets
// per sample (where a set's source lies inside the tile, after that set's own hand-over); the gains that ramp are stepped serially (the running sum is not a closed form in
// floating point) by the wavefront's own lanes, one per gain, into rows of the late half that are idle in S5.
// SF (the FP plain and HY builds, mono / stereo, single-slot batches): the send shelf filters inside (reference apply_filters,
// src/oalsfxpp.cpp:3101-3143, at most two biquads per send and input channel) for the instances that have one switched on, instead of
// the pre-pass kernel and its planes.  The tile loop is skewed once more: iteration it filters the frame of tile it + 1 -- feed-forward
// sums with lane = frame in S1 and S3, the two recurrences on 4 x (sends x channels) lanes of the wavefronts that idle in S2 and S4 --
// beside the input half of tile it, which takes the filtered rows instead of the raw frame, and the late half of tile it - 1.  No
// barrier is added; a workgroup with such an instance runs one iteration more.
// RG (a variant of the ST build): calls that are not a whole number of tiles.  The last tile holds L < 64 samples: its lanes
// from L on compute along but store nothing, the recurrences and the modulation smoother stop at L, and the histories are
// taken from sample L - 1.

// LDS of one workgroup of the cooperative kernel
// SF: the send filters inside (the FP plain and HY builds of single-slot batches): per wave two rows per send and input channel -- the
// first shelf's feed-forward sums / output, the second shelf's -- and 64 floats of coefficients, edge samples and histories
namespace sfm { enum { TAB = 0 /* [send][12] */, EDGE = 24 /* [channel][4]: samples 0, 1, 62, 63 of the tile being filtered */, HIST = 32 /* [row][6] */, SIZE = 64 }; }

template <int CH, int NW, bool FP = false, bool MD = true, bool ST = true, bool SF = false, int CR = 0>
struct SteadyShared {
    static constexpr bool MC = CH > 2;
    static constexpr int kMcBase = ut::SIZE + 8 * kRow; // multichannel tables behind the hand-over rows (the modulation row shares the first of
                                                        // them): GOUT8 [8 stage-lines][8 channels], GDIR8 [8][8], GAUX8 [8][4]
    // FP (proven-steady instances only, nothing falls back inside): the rows, the table, the record's odds and ends, and what the
    // build's own extras need (modulation row, hand-over rows)
    static constexpr int kFpMisc = ut::SIZE + (ST ? 8 * kRow : MD ? 64 : 0);
    // mono / stereo otherwise: sized for the general path, which non-steady instances fall back to
    static constexpr int kSteadyFloats = kSteadyGroups * 4 * kRow + ut::SIZE + 8 * kRow; // rows, table, hand-over rows (the modulation row is the first of them)
    static constexpr int kFloats = FP ? kSteadyGroups * 4 * kRow + kFpMisc + 64
                                 : MC ? kSteadyGroups * 4 * kRow + kMcBase + 160
                                      : (Lds<CH>::kFloats > kSteadyFloats ? Lds<CH>::kFloats : kSteadyFloats);
    alignas(16) float lds_all[NW][kFloats];
    float chain_all[NW][4][coop::SIZE]; // [wave][line]: filter histories and feedback coefficients
    // (the workgroup's LDS decides how many fit a CU -- four at 40 KiB each --, and the multichannel builds sit just under that: what a
    // build does not use is kept to a stub)
    unsigned tapn_all[(FP || MC) ? 1 : NW][24]; // XF: [wave][group * 4 + line]: the taps being faded in, as byte distances like ut::TAP4
    alignas(16) float sf_rows[SF ? NW : 1][SF ? 4 * CH : 1][SF ? kRow : 4]; // SF: [wave][stage * 2 * CH + send * CH + channel]
    float sf_misc[SF ? NW : 1][SF ? sfm::SIZE : 4];
    alignas(16) float cr_tails[CR == 2 ? NW : 1][CR == 2 ? 5 : 1][CR == 2 ? 32 : 1][4]; // CR == 2: [wave][store site][lane][line]: the tile's samples past the last line boundary
    int sf_all[NW]; // SF: which instances of the group filter their sends in here
    int go_all[NW];
    int eax_all[NW]; // which instances of the group are EAX reverbs (second input shelf)
};

// The work of workgroup `group` of the cooperative kernel (its own kernel below; also one half of k_slot_mixed).
// FP (with any of plain / HY / MD / ST, whole tiles, mono / stereo): the launch holds only instances the host has *proven* steady
// (the device reported them settled and at rest, DESIGN 4, and nothing has been uploaded for them since).  There is no steady-state
// test to fail and no general path to fall back to -- neither its registers, its scratch frame nor its 16 KiB of gain-ramp rows --
// and a buffer starts from the instance's hot record (namespace hot): one 16-byte load per lane instead of a tree of descriptor
// loads.  A record whose stamp does not match is rebuilt from the descriptors (first call after a promotion, or after another
// kernel advanced the instance); an instance that then fails the steady-state test after all is counted in ctx.fault and left alone.
// CR (the plain FP builds: every tap three tiles away, early taps and late-line offsets 32 samples more -- kPlainMinTapAhead): ring stores
// that cover whole 128-byte lines.  A tile's 64 samples of a ring line are 256 contiguous bytes, but where they start is the delay
// line's write position: after a call that was not a multiple of 32 frames every such store begins and ends inside a line, for the rest
// of the instance's life, and the late feed's stores (main delay, late_feed_tap samples back: 16012 with the defaults, 12 past a line)
// never were aligned.  Memory takes a line written in two parts badly (what launches hand on is uncached: no L2 merges the parts):
// 256-frame calls after one call of 441, 100 or 37 frames took 13 - 21 % longer (profiles/r03n_round3_end/misaligned_ring_positions.txt),
// and with the stores rounded down to lines, an ablation, nothing of that was left (profiles/r04a_line_aligned_stores/).  So a store
// site holds back the r samples of its tile that lie past the last line boundary, r = position % 32, and writes them with the next
// tile's: lane L stores the sample r places before its own (a lane rotation), lanes below r the ones held back -- in a register per
// value (CR >= 1: the late feed's four, every plain FP build) or in 32 floats of LDS per value (CR == 2: all six sites, the build the
// host picks when the write position of an instance of the launch is off the grid).  The call's first tile leaves the lanes below r
// alone (the call before wrote those samples), its last tile writes its held-back samples as well: two partial lines per ring line and
// call instead of two per tile.  (Measured and dropped: reading the r samples in front of the first tile again, so that the call's
// first store covers whole lines too -- the wait for them in front of the loop cost more than the partial lines, 45.2 against 44.2 us;
// and the late feed alone held back in the build for positions on the grid, CR == 1: 41.4 against 40.5 us with the reload, no
// difference without.)  Loads that reach into the tile before are the reason for kPlainMinTapAhead: a held-back sample is in
// memory one tile later.
template <int CH, int NW, bool TL = false, bool HY = false, bool MD = false, bool ST = false, bool RG = false, bool FP = false, bool XF = false, bool NF = false, bool SF = false, int CR = 0, class SH>
__device__ __forceinline__ void reverb_steady_group(const KernelCtx& ctx, int slot, const int* __restrict__ list, int count, int flags, const int group,
                                                    SH& sh)
{
    static_assert(!FP || CH <= 2, "the proven-steady builds: mono / stereo");
    static_assert(!(FP && RG) || !SF, "the proven ragged builds: the plain (with or without line-aligned stores) and the most general kind");
    static_assert(!XF || (CH <= 2 && !RG && !FP && HY && MD && ST), "the cross-fading build: a variant of the most general one, mono / stereo, whole tiles");
    static_assert(!SF || (FP && !MD && !ST && NW == 4), "send filters inside: the FP plain and HY builds");
    constexpr int kSfRows = 2 * CH; // SF: rows per stage = sends (direct, this slot's auxiliary) x input channels
    // XF: dword offset (from the table) of the taps being faded in: a second tap table with ut::TAP4's layout, in the workgroup's own array
    const int kTapN = static_cast<int>(reinterpret_cast<float*>(&sh.tapn_all[(FP || CH > 2) ? 0 : wib_of(threadIdx.x)][0]) - (sh.lds_all[wib_of(threadIdx.x)] + kSteadyGroups * 4 * kRow));
    // TL: measurement build, every 64th workgroup stamps the shader clock at each phase boundary (up to 96 stamps per wave)
    int ts_i = 0;
    auto stamp = [&]() {
        if (TL && (group & 63) == 0 && (threadIdx.x & 63) == 0 && ts_i < 96)
            ctx.timeline[((group >> 6) * NW + (threadIdx.x >> 6)) * 96 + ts_i++] = clock64();
    };
    stamp();
    static_assert(CH <= 2 || CH == 8, "mono, stereo, or the multichannel build");
    constexpr bool MC = CH > 2;
    constexpr int kMcBase = SH::kMcBase;
    const int nch = MC ? ctx.channels : CH;
    auto& lds_all = sh.lds_all;
    auto& chain_all = sh.chain_all;
    auto& go_all = sh.go_all;
    auto& eax_all = sh.eax_all;

    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int w = group * NW + wib;
    bool valid = w < count; // (cleared when the instance's turn in a run of chained launches never came: the wavefront then only keeps its workgroup company)
    const int lane = threadIdx.x & 63;
    const int frames = ctx.frames;
    const bool first = (flags & kFirst) != 0;
    const bool last = (flags & kLast) != 0;

    float* lds = lds_all[wib];
    auto rowI = [&](int group, int c) -> float* { return lds + (group * 4 + c) * kRow; };       // input half: send mix, shelves
    auto rowL = [&](int group, int c) -> float* { return lds + ((3 + group) * 4 + c) * kRow; }; // late half: T60 sections
    float* utf = lds + kSteadyGroups * 4 * kRow;
    unsigned* utu = reinterpret_cast<unsigned*>(utf);

    // (the idle wavefronts of an incomplete workgroup run beside the workgroup's own first instance: they take part in its barriers and
    // chain phases and need some instance's records to read -- one that this workgroup waits for anyway, see the wait below)
    const int wa = valid ? w : group * NW;
    const int inst = (FP && ctx.list_first >= 0) ? ctx.list_first + wa : __builtin_amdgcn_readfirstlane(list[wa]);
    const size_t sidx = static_cast<size_t>(inst) * ctx.slots + slot;
    typedef const __attribute__((address_space(4))) oalsfx_slot_params ConstSlotParams;
    ConstSlotParams& SP = *(ConstSlotParams*)(uintptr_t)(ctx.params + sidx);
    const auto& P = SP.u.reverb;
    oalsfx_slot_state& SS = ctx.state[sidx];
    oalsfx_reverb_state& S = SS.u.reverb;
    GlobalBytes* slab_b = (GlobalBytes*)(uintptr_t)ctx.rings[sidx];

    // Chained launches: the launch before this one runs on another stream and may still be at work; this instance's turn comes
    // when that launch is through with it.
    //  - Everything one launch hands to the next -- delay lines, state, hot records, send-filter histories, this word -- lives in memory
    //    the L2s do not cache (hipDeviceMallocUncached, batch.cpp): what a wavefront stored is in memory, for every XCD to see, once its
    //    stores are acknowledged (s_waitcnt vmcnt(0), then the word).  (Agent-scope fences over cached memory -- a write-back of the L2
    //    per wavefront -- were measured first: 245 us per step instead of 50.)
    //  - No cache line holds bytes of two instances (SlotStateLines, common.hpp), and nobody reads the instance's lines in this launch
    //    before its turn has come (the idle wavefronts of an incomplete workgroup, which run beside the workgroup's first instance, wait
    //    for that instance's turn as well -- when they ran beside the first instance of their kind's list without waiting, that CU's
    //    L1 held old lines of it: found with 70 instances, tests/test_gpu_chained.py): this CU's vector L1, emptied when the launch
    //    started, holds none of them before the wait is over, and needs no invalidate behind it (an agent-scope acquire there
    //    -- buffer_inv sc1 -- cost 9 to 12 us per step, more than the overlap gains; profiles/r03k_chained_launches).  The scalar cache
    //    is another matter: it is not written through by vector stores, so a line the launch before loaded through it may still be
    //    there; s_dcache_inv costs nothing measurable.
    //  - The wait ends: the launch before has its workgroups on the chip before this one gets its first (batch.cpp, k_chain_gate), and
    //    a count-out reports through the fault word rather than hang.  An instance whose turn did not come is left alone -- nothing of
    //    it is read into the tile loop, nothing written, its word stays as it is, so that the launches behind count out on it too
    //    instead of working on what this one skipped -- and the host, which sees the fault word at its next synchronising call, fails
    //    that call and every later one of the batch (check_fault, batch.cpp).
    unsigned cu_before = 0; // the CU the launch before ran this instance on (0: this launch is a run's first)
    // (FP builds ask for the two CU names when the turn has come and look at them beside the hot record's load, one round trip for both
    // instead of two in a row: OALSFX_CU_CHECK_BESIDE_RECORD)
    unsigned v_cu1 = 0, v_cu2 = 0;
    bool cu_check_pending = false;
    const int dbg = flags >> 8; // test switches of the hand-over, below
    if (ctx.turn_started != nullptr && threadIdx.x == 0) __hip_atomic_fetch_add(ctx.turn_started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ctx.turn != nullptr && ctx.turn_wait != 0u) {
        // Test switches (OALSFX_DEBUG_FLAGS 1 / 2 / 4, tests/test_gpu_chained.py): 1 every wavefront pays for the agent-scope acquire behind
        // its wait, as if it always ran where the launch before did (the same-CU path, taken by none of 4.9 million hand-overs on a full
        // chip: exercised deterministically); 2 none does; 4 before its turn has come the wavefront reads the instance's hot record,
        // state and the all-pass rings' lines it is about to use -- what the design's invariant "nobody reads an instance's lines before
        // its turn" forbids -- so that this CU's L1 holds them as they were while the launch before is still writing them: with 2 the
        // results must come out wrong (the negative control), with 1 the acquire must put them right.
        if (dbg & 4) {
            const unsigned* rec = ctx.hot + sidx * hot::SIZE;
            const unsigned* st = reinterpret_cast<const unsigned*>(ctx.state + sidx);
            unsigned junk = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned a;
                asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(rec + 64 * k + lane) : "memory"); // (the wait inside: the compiler does not know the load is in flight)
                junk ^= a;
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                unsigned a;
                asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(st + 64 * k + lane) : "memory");
                junk ^= a;
            }
            if (slab_b != nullptr) {
                // (the two all-pass rings: 512 + 1024 frames x 4 lines at 48 kHz, placed behind the three long rings; a dword of every line)
                const unsigned* ring = reinterpret_cast<const unsigned*>(ctx.rings[sidx]);
                const oalsfx_reverb_params& PGd = ctx.params[sidx].u.reverb;
                for (int r = 0; r < 5; ++r) {
                    if (r == OALSFX_RV_MAIN || r == OALSFX_RV_EARLY_LINE || r == OALSFX_RV_LATE_LINE) continue;
                    const int words = 4 * PGd.ring_len[r];
                    for (int w0 = 0; w0 < words; w0 += 64 * 32) {
                        unsigned a;
                        const int at = min(w0 + 32 * lane, words - 1);
                        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(ring + PGd.ring_off[r] + at) : "memory");
                        junk ^= a;
                    }
                }
            }
            if (!(flags & kFirst) && ctx.mixbuf != nullptr) {
                // (a step of two launches: the mix of the slots in front, which the ring-light kernel's launch is writing -- a dword of every line)
                const unsigned* mb = reinterpret_cast<const unsigned*>(ctx.mixbuf + static_cast<size_t>(inst) * nch * OALSFX_MAX_CHUNK);
                for (int c = 0; c < nch; ++c) {
                    unsigned a;
                    asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(mb + c * OALSFX_MAX_CHUNK + 32 * lane) : "memory");
                    junk ^= a;
                }
            }
            if (junk == 0x7E57AB1Eu && ctx.timeline) ctx.timeline[0] = junk; // (keeps the loads)
        }
        int lost = 0;
        if (lane == 0 && !(OALSFX_CHAIN_EXP & 8)) {
            unsigned spins = 0;
            while (__hip_atomic_load(ctx.turn + sidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ctx.turn_wait) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 20)) {
                    if (ctx.fault && valid) __hip_atomic_fetch_add(ctx.fault, kFaultTurn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    lost = 1;
                    break;
                }
            }
        }
        if (__builtin_amdgcn_readfirstlane(lost)) valid = false;
#if OALSFX_CHAIN_EXP & 1
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#else
        // Readers of the instance's lines are left that this launch's start did not come after: the instance's own wavefronts of the
        // launches still in flight when this one started -- the launch before, and with three launches in flight the one before that --
        // which kept reading them while this launch was already on the chip (that is the overlap); the lines have been written since (by
        // those wavefronts themselves: an all-pass ring comes round within a call or two; by the launch in between).  A CU that one of
        // them ran on may hold such a line as it was read.  So each launch leaves the CU's name beside the word, and the name the launch
        // before it left one further on: where this wavefront runs on one of the two, it does pay for the agent-scope acquire
        // (buffer_inv sc1: this CU's L1 dropped).  Rare on a full chip (a workgroup is on the chip, waiting, before the one it waits for
        // leaves its CU); with few workgroups, whose places shift from launch to launch as instances change kind, it is what the random
        // runs of tests/test_gpu_chained.py found: 35 of 6000 wrong with the launch before alone looked at, three launches in flight.
        if constexpr (FP && OALSFX_CU_CHECK_BESIDE_RECORD) {
            if (lane == 0) {
                v_cu1 = __hip_atomic_load(ctx.turn_cu + sidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v_cu2 = __hip_atomic_load(ctx.turn_cu2 + sidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            cu_check_pending = true;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // (no instruction: keeps the loads below behind the wait)
        } else {
            unsigned before_cu = 0, before_that_cu = 0;
            if (lane == 0) {
                before_cu = __hip_atomic_load(ctx.turn_cu + sidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                before_that_cu = __hip_atomic_load(ctx.turn_cu2 + sidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            before_cu = __builtin_amdgcn_readfirstlane(before_cu);
            before_that_cu = __builtin_amdgcn_readfirstlane(before_that_cu);
            cu_before = before_cu;
            // (the launch before the last counts only where it may have been at work when this launch started: the host knows -- this
            // launch sits behind it in its stream, or it does not)
            if (!(dbg & 2) && ((dbg & 1) || before_cu == this_cu() || (ctx.turn_two_back != 0u && before_that_cu == this_cu()))) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (ctx.turn_started != nullptr && lane == 0) __hip_atomic_fetch_add(ctx.turn_started + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (a count for the records)
            } else {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // (no instruction: keeps the loads below behind the wait)
            }
        }
#endif
        __builtin_amdgcn_s_dcache_inv();
        if (FP && OALSFX_CU_CHECK_BESIDE_RECORD && !(OALSFX_CHAIN_EXP & 1)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // (the CU names stay in flight)
        else __builtin_amdgcn_s_waitcnt(0);
    }
    const int l4 = lane & 3;
    const int q_stage = lane / (4 * CH), q_line = (lane / CH) & 3, q_chan = lane % CH;
    const bool q_valid = lane < 8 * CH && q_chan < nch;
    // what the tile loop and the epilogue need to know about the instance (wave-uniform), from the hot record or the descriptors
    bool go = false, eax = false, mod_on = false, has_filter = false;
    unsigned late_mask = 0;  // hybrid build: tap groups requested in their own tile
    unsigned short_mask = 0; // ST build: tap groups with a source inside the tile
    unsigned long long aud_dir = 0, aud_aux = 0, aud_out = 0; // which gains are audible (bit layout as the tables)
    int offset = 0, v_modidx = 0, v_modrange = 1;
    float mod_f = 0.0F, mod_depth = 0.0F, mod_coeff = 0.0F;
    unsigned send_mask = 0; // FP: sends whose filter histories follow the input (bit 0 direct, bit 1 + s the send to slot s)
    unsigned epoch_now = 0;
    unsigned* miscu = reinterpret_cast<unsigned*>(utf + (FP ? SH::kFpMisc : 0)); // FP: image of the record's MISC block
    bool hit = false;
    // XF: an instance that folds in a property change, cross-fades or ramps gains in this call
    bool xf_active = false;
    int fc0 = OALSFX_RV_FADE_SAMPLES; // its fade count at the start of the call (after a pending change has been folded in)
    float g_cur = 0.0F, g_tgt = 0.0F;  // lane q < 8 * CH: an output gain and its target
    float early_in0 = 0.0F, early_in1 = 0.0F; // FP: the first tile's frame, requested beside the hot record
    float hist_new[2] = {0.0F, 0.0F}, hist_old[2] = {0.0F, 0.0F}; // FP: the call's last two frames per channel, for the send filters' histories
    if constexpr (FP) {
        // ---- the hot record: 1 KiB, one 16-byte load per lane, straight into the LDS tables ----
        const v4u* rec = reinterpret_cast<const v4u*>(ctx.hot + sidx * hot::SIZE);
        const v4u r = rec[lane];
        // the first tile's frame does not depend on the record (unless the send-filter pre-pass ran): it travels beside it
        if (!(flags & kFiltered)) {
            const float* raw = ctx.raw_src + static_cast<size_t>(inst) * ctx.io_stride;
            const int fl = RG ? min(lane, frames - 1) : lane; // (a ragged call may be shorter than a tile)
            if (CH == 2) {
                if (OALSFX_NT & 32) {
                    const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(raw + static_cast<size_t>(fl) * 2));
                    early_in0 = v.x; early_in1 = v.y;
                } else {
                    const float2 v = *reinterpret_cast<const float2*>(raw + static_cast<size_t>(fl) * 2);
                    early_in0 = v.x; early_in1 = v.y;
                }
            } else {
                early_in0 = raw[fl];
            }
        }
        unsigned v_epoch = ctx.inst_epoch[inst];
        int v_off = S.offset;
        v4u rr = r;
        if (cu_check_pending) {
            // the CU names asked for when the turn came (see the wait above): where this wavefront runs on the CU of the launch before --
            // or of the one before that, where it may still have been at work -- this CU's L1 may hold the instance's lines as that launch
            // read them: dropped (agent-scope acquire), and what was just loaded through it asked for again
            const unsigned before_cu = __builtin_amdgcn_readfirstlane(v_cu1), before_that_cu = __builtin_amdgcn_readfirstlane(v_cu2);
            cu_before = before_cu;
            if (!(dbg & 2) && ((dbg & 1) || before_cu == this_cu() || (ctx.turn_two_back != 0u && before_that_cu == this_cu()))) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (ctx.turn_started != nullptr && lane == 0) __hip_atomic_fetch_add(ctx.turn_started + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (a count for the records)
                rr = rec[lane];
                v_off = S.offset;
            }
        }
        epoch_now = __builtin_amdgcn_readfirstlane(v_epoch);
        const int offset_now = __builtin_amdgcn_readfirstlane(v_off);
        if (lane < 32) *reinterpret_cast<v4u*>(utu + 4 * lane) = rr;
        else if (lane < 48) *reinterpret_cast<v4u*>(&chain_all[wib][0][0] + 4 * (lane - 32)) = rr;
        else *reinterpret_cast<v4u*>(miscu + 4 * (lane - 48)) = rr;
        wave_sync();
        hit = valid && __builtin_amdgcn_readfirstlane(miscu[hot::M_EPOCH]) == epoch_now &&
              static_cast<int>(__builtin_amdgcn_readfirstlane(miscu[hot::M_OFFSET])) == offset_now;
        if (hit) {
            go = true;
            offset = offset_now;
            eax = __builtin_amdgcn_readfirstlane(miscu[hot::M_EAX]) != 0;
            has_filter = __builtin_amdgcn_readfirstlane(miscu[hot::M_HAS_FILTER]) != 0;
            aud_dir = __builtin_amdgcn_readfirstlane(miscu[hot::M_AUD_DIR]);
            aud_aux = __builtin_amdgcn_readfirstlane(miscu[hot::M_AUD_AUX]);
            aud_out = __builtin_amdgcn_readfirstlane(miscu[hot::M_AUD_OUT]);
            if (HY) late_mask = __builtin_amdgcn_readfirstlane(miscu[hot::M_LATE_MASK]);
            if (ST) short_mask = __builtin_amdgcn_readfirstlane(miscu[hot::M_SHORT_MASK]);
            if (MD) {
                mod_on = __builtin_amdgcn_readfirstlane(miscu[hot::M_MOD_ON]) != 0;
                mod_f = __uint_as_float(__builtin_amdgcn_readfirstlane(miscu[hot::M_MOD_F]));
                v_modrange = static_cast<int>(__builtin_amdgcn_readfirstlane(miscu[hot::M_MOD_RANGE]));
                mod_depth = __uint_as_float(__builtin_amdgcn_readfirstlane(miscu[hot::M_MOD_DEPTH]));
                mod_coeff = __uint_as_float(__builtin_amdgcn_readfirstlane(miscu[hot::M_MOD_COEFF]));
            }
            {
                v_modidx = static_cast<int>(__builtin_amdgcn_readfirstlane(miscu[hot::M_MOD_INDEX])); // advances in every build
                if (!MD) v_modrange = static_cast<int>(__builtin_amdgcn_readfirstlane(miscu[hot::M_MOD_RANGE]));
            }
            send_mask = __builtin_amdgcn_readfirstlane(miscu[hot::M_SEND_MASK]);
            if (lane == 0) {
                go_all[wib] = 1;
                eax_all[wib] = eax ? 1 : 0;
            }
        }
    }
    if (!hit) {
    // ---- everything the steady-state path needs from the descriptors, requested in one go: the loads below do not
    // depend on each other, so the prologue costs one memory round trip after the list entry instead of three
    const oalsfx_reverb_params& PG = ctx.params[sidx].u.reverb; // per-lane (vector) reads of the parameter block
    const oalsfx_source_params& SG = ctx.source[inst];
    const unsigned v_seen = SS.seen_seq;
    const int v_fade = S.fade_count;
    const float v_modf = S.mod_filter;
    const int v_offset = S.offset;
    v_modidx = S.mod_index; v_modrange = S.mod_range;
    const int v_tap = (&S.cur_early_tap[0])[min(lane, 23)];
    int v_newtap = 0; // XF: the taps a change would fade in (the six arrays are not adjacent in the parameter block)
    if (XF) {
        const int g = min(lane >> 2, 5);
        const int32_t* from = g == 0 ? PG.early_tap : g == 1 ? PG.early_ap_off : g == 2 ? PG.early_line_off : g == 3 ? PG.late_tap : g == 4 ? PG.late_ap_off : PG.late_line_off;
        v_newtap = from[l4];
    }
    const int v_ring_off = PG.ring_off[min(lane >> 2, 4)], v_ring_len = PG.ring_len[min(lane >> 2, 4)];
    const int v_ring_len5 = PG.ring_len[min(lane, 4)];
    const float v_gcur = q_valid ? (q_stage ? S.late_cur_gain[q_line][q_chan] : S.early_cur_gain[q_line][q_chan]) : 0.0F;
    const float v_gtgt = q_valid ? (q_stage ? PG.late_pan[q_line][q_chan] : PG.early_pan[q_line][q_chan]) : 0.0F;
    const float v_ecoef = PG.early_tap_coeff[l4], v_elcoef = PG.early_line_coeff[l4];
    const float v_tl0 = PG.t60_lf[l4][0], v_tl1 = PG.t60_lf[l4][1], v_tl2 = PG.t60_lf[l4][2];
    const float v_th0 = PG.t60_hf[l4][0], v_th1 = PG.t60_hf[l4][1], v_th2 = PG.t60_hf[l4][2];
    const float v_tmid = PG.t60_mid[l4];
    const float v_lpx0 = S.lp[l4].x[0], v_lpx1 = S.lp[l4].x[1], v_lpy0 = S.lp[l4].y[0], v_lpy1 = S.lp[l4].y[1];
    const float v_hpy0 = S.hp[l4].y[0], v_hpy1 = S.hp[l4].y[1];
    const float v_t60x = S.t60[l4][0][0], v_t60o1 = S.t60[l4][0][1], v_t60o2 = S.t60[l4][1][1];
    // send gains: mono / stereo lane = c * 2 + o (direct), c * 4 + k (aux); multichannel lane = c * 8 + o, c * 4 + k
    const float v_gdir = MC ? SG.direct.gains[lane >> 3][lane & 7] : SG.direct.gains[(lane >> 1) & 1][lane & 1];
    const float v_gaux = MC ? SG.aux[slot].gains[(lane >> 2) & 7][lane & 3] : SG.aux[slot].gains[(lane >> 2) & 1][lane & 3];
    mod_depth = P.mod_depth; mod_coeff = P.mod_coeff;
    mod_f = v_modf;
    bool xf_pending = false;
    if constexpr (XF) {
        xf_pending = v_seen != SP.update_seq;
        fc0 = v_fade;
        if (xf_pending) {
            // what the reference's update does to the state (src/oalsfxpp.cpp:7028-7031, 6062-6075): the modulator's index follows its
            // new range, and taps that moved start a cross-fade
            v_modidx = static_cast<int>(static_cast<long long>(v_modidx) * P.mod_range / v_modrange);
            v_modrange = P.mod_range;
            if (__ballot(lane < 24 && v_newtap != v_tap) != 0ULL) fc0 = 0;
        }
    }
    const bool fading = XF && fc0 < OALSFX_RV_FADE_SAMPLES;

    // ---- is this instance in its steady state for the whole buffer?  (XF: or in a state the cross-fading tiles handle) ----
    go = valid && (RG || (frames & 63) == 0) && (XF ? (!fading || (fc0 & 63) == 0) : ((v_seen == SP.update_seq) && (v_fade >= OALSFX_RV_FADE_SAMPLES))) &&
         (MD || ((P.mod_depth == 0.0F) && (v_modf == 0.0F)));
    mod_on = MD && ((P.mod_depth != 0.0F) || (v_modf != 0.0F));
    g_cur = v_gcur; g_tgt = v_gtgt;
    {
        // the last chunk of the buffer has the smallest ramp counter, hence the largest step: no ramp there, no ramp anywhere
        const int last_chunk = frames - ((frames - 1) / OALSFX_RV_MAX_UPDATE) * OALSFX_RV_MAX_UPDATE;
        const float step = (v_gtgt - g_cur) * (1.0F / static_cast<float>(last_chunk));
        const bool ramping = __ballot(q_valid && fabsf(step) > FLT_EPSILON) != 0ULL;
        if (ramping && !XF) go = false;
        xf_active = XF && (xf_pending || fading || ramping); // (a fade splits the call into other blocks than these: the XF tiles step block by block)
        // every tap at least two tiles away from its write position (late taps: from the late feed position); the hybrid
        // build accepts one tile and requests the groups that are closer than two at the top of their own tile
        const unsigned tp = (lane < 24) ? 4u * static_cast<unsigned>(v_tap) : 0xFFFFFFFFu;
        const unsigned tpn = (fading && lane < 24) ? 4u * static_cast<unsigned>(v_newtap) : 0xFFFFFFFFu; // XF: the taps being faded in must keep their distance too
        unsigned feed4 = (lane >> 2) == 3 ? 4u * static_cast<unsigned>(P.late_feed_tap) : 0u;
        // a modulated late line reads up to |depth| samples closer (the smoother moves monotonically towards the depth)
        if (MD && (lane >> 2) == 5) feed4 = 4u * (1u + static_cast<unsigned>(fmaxf(fabsf(P.mod_depth), fabsf(v_modf))));
        const int grp = lane >> 2;
        // shortest distance accepted per group (ST build only)
        // (ST: early and late taps and the early line of any length, all-pass offsets from four samples -- up to fifteen evaluations ahead
        // per tile --, the late line a whole tile: its loop runs through the T60 chain phases)
        const unsigned shortest = !ST ? (HY ? 256u : (FP ? 4u * ((grp == 0 || grp == 5) ? kPlainMinTapAhead : kPlainMinTap) : 512u)) : (grp == 0 || grp == 2 || grp == 3) ? 0u : (grp == 1 || grp == 4) ? 16u : 256u;
        if (__ballot(tp >= shortest + feed4 && tpn >= shortest + feed4) != ~0ULL) go = false;
        if (ST) {
            const unsigned long long in_tile = __ballot(lane < 24 && (tp < 256u + feed4 || tpn < 256u + feed4));
#pragma unroll
            for (int g = 0; g < 6; ++g) short_mask |= ((in_tile >> (4 * g)) & 0xFULL) ? 1u << g : 0u;
        }
        if (HY) {
            const unsigned long long close = __ballot(tp < 512u + feed4 || tpn < 512u + feed4);
#pragma unroll
            for (int g = 0; g < 6; ++g) late_mask |= ((close >> (4 * g)) & 0xFULL) ? 1u << g : 0u;
        }
        // At rest?  From which block length on no output gain would be ramped (the test above depends on the size of this call) --
        // what the host needs to know before it may list the instance for an FP build.  (The vote is taken by the whole wavefront,
        // outside the lane-0 branch.)
        if (!FP && ctx.exact) {
            const unsigned level = gains_rest_level(q_valid, g_cur, v_gtgt);
            if (valid && go && lane == 0 && !xf_active) ctx.exact[sidx] = level; // (an instance in transition reports at the end of the call)
        }
    }
    eax = P.is_eax != 0; // plain reverb and EAX reverb instances may share a workgroup
    has_filter = (FP || (flags & kFiltered) != 0) && instance_has_send_filter(ctx, inst);
    if (lane == 0) {
        go_all[wib] = go ? 1 : 0;
        eax_all[wib] = (go && eax) ? 1 : 0;
    }
    if (go) {
        // ---- per-wave table of instance constants in LDS (see namespace ut), chain data per line ----
        // (flags >> 8) & 32 / 64: timing experiment only (OALSFX_DEBUG_FLAGS), taps rounded to 128 / 256 bytes, results wrong
        if (lane < 24) utu[ut::TAP4 + lane] = (4u * static_cast<unsigned>(v_tap)) & ((flags & (64 << 8)) ? ~255u : (flags & (32 << 8)) ? ~127u : ~0u);
        if (XF && lane < 24) utu[kTapN + lane] = 4u * static_cast<unsigned>(v_newtap);
        if (lane < 20) utu[ut::LO + lane] = static_cast<unsigned>(v_ring_off + l4 * v_ring_len) << 2;
        if (lane < 5) utu[ut::BMASK + lane] = static_cast<unsigned>(v_ring_len5 - 1) << 2;
        if (lane < 4) {
            utf[ut::ECOEF + lane] = v_ecoef;
            utf[ut::ELCOEF + lane] = v_elcoef;
            utf[ut::TL0 + lane] = v_tl0;
            utf[ut::TL1 + lane] = v_tl1;
            utf[ut::TH0 + lane] = v_th0;
            utf[ut::TH1 + lane] = v_th1;
            float* ch = chain_all[wib][lane];
            ch[coop::LPX0] = v_lpx0; ch[coop::LPX1] = v_lpx1;
            ch[coop::LPY0] = v_lpy0; ch[coop::LPY1] = v_lpy1;
            ch[coop::HPY0] = v_hpy0; ch[coop::HPY1] = v_hpy1;
            ch[coop::T60X] = v_t60x; ch[coop::T60O1] = v_t60o1; ch[coop::T60O2] = v_t60o2;
            ch[coop::LP_A1] = P.lp.a1; ch[coop::LP_A2] = P.lp.a2; ch[coop::HP_A1] = P.hp.a1; ch[coop::HP_A2] = P.hp.a2;
            ch[coop::T_L2] = v_tl2; ch[coop::T_H2] = v_th2; ch[coop::T_MID] = v_tmid;
            if (!MC) utf[ut::GDIR + lane] = v_gdir;
        }
        if (!MC && lane < 8) utf[ut::GAUX + lane] = v_gaux;
        if (MC) {
            utf[kMcBase + lane] = q_valid ? g_cur : 0.0F;
            utf[kMcBase + 64 + lane] = v_gdir;
            if (lane < 32) utf[kMcBase + 128 + lane] = v_gaux;
        }
        if (lane == 0) {
            utu[ut::FEED4] = 4u * static_cast<unsigned>(P.late_feed_tap);
            utf[ut::MISC + 0] = P.density_gain; utf[ut::MISC + 1] = P.ap_feed_coeff; utf[ut::MISC + 2] = P.mix_x; utf[ut::MISC + 3] = P.mix_y;
            utf[ut::LPB + 0] = P.lp.b0; utf[ut::LPB + 1] = P.lp.b1; utf[ut::LPB + 2] = P.lp.b2; utf[ut::LPB + 3] = 0.0F;
            utf[ut::HPB + 0] = P.hp.b0; utf[ut::HPB + 1] = P.hp.b1; utf[ut::HPB + 2] = P.hp.b2; utf[ut::HPB + 3] = 0.0F;
        }
        if (!MC && q_valid) utf[ut::GOUT + (CH == 1 ? 2 * lane : lane)] = g_cur;
        if (MC) {
            aud_dir = __ballot((lane >> 3) < nch && (lane & 7) < nch && audible(v_gdir));  // bit c * 8 + o
            aud_aux = __ballot(lane < 32 && (lane >> 2) < nch && audible(v_gaux));          // bit c * 4 + k
        } else {
            aud_dir = __ballot(lane < 4 && audible(v_gdir));   // bit c * 2 + o
            aud_aux = __ballot(lane < 8 && audible(v_gaux));   // bit c * 4 + k
        }
        if (CH == 1) { aud_dir &= 1u; aud_aux &= 0xFu; }
        aud_out = __ballot(q_valid && audible(g_cur));        // multichannel: bit (stage * 4 + line) * 8 + channel
        if (CH == 1) aud_out = ((aud_out & 1u) | ((aud_out & 2u) << 1) | ((aud_out & 4u) << 2) | ((aud_out & 8u) << 3) | ((aud_out & 16u) << 4) |
                                ((aud_out & 32u) << 5) | ((aud_out & 64u) << 6) | ((aud_out & 128u) << 7));
        offset = v_offset;
        if constexpr (FP) {
            // the parts of the record that do not change from call to call; the epilogue adds the rest
            typedef const __attribute__((address_space(4))) oalsfx_source_params ConstSourceParams;
            ConstSourceParams& CS = *(ConstSourceParams*)(uintptr_t)(ctx.source + inst);
            send_mask = 1u;
            for (int k = 0; k < ctx.slots; ++k)
                if (CS.aux[k].out_channels != 0) send_mask |= 2u << k;
            if (lane == 0) {
                miscu[hot::M_EAX] = eax ? 1u : 0u;
                miscu[hot::M_HAS_FILTER] = has_filter ? 1u : 0u;
                miscu[hot::M_AUD_DIR] = static_cast<unsigned>(aud_dir);
                miscu[hot::M_AUD_AUX] = static_cast<unsigned>(aud_aux);
                miscu[hot::M_AUD_OUT] = static_cast<unsigned>(aud_out);
                miscu[hot::M_LATE_MASK] = late_mask;
                miscu[hot::M_SHORT_MASK] = short_mask;
                miscu[hot::M_MOD_RANGE] = static_cast<unsigned>(v_modrange);
                miscu[hot::M_MOD_DEPTH] = __float_as_uint(mod_depth);
                miscu[hot::M_MOD_COEFF] = __float_as_uint(mod_coeff);
                miscu[hot::M_SEND_MASK] = send_mask;
            }
        }
        wave_sync(); // this wave's table is complete: the first tile's requests below read it
    } else if (FP && valid && lane == 0) {
        // the host listed an instance that is not steady: reported by the next synchronising call (the counter lives in host memory)
        __hip_atomic_fetch_add(ctx.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    } // !hit
    stamp(); // [1] hot record or descriptors read, steady-state test done, tables written
    // SF: this instance filters its sends in here (the host says so for the launch: single-slot batch, every such instance on an SF build)
    const bool sf = SF && go && first && (flags & kFilterInside) != 0 && has_filter;
    float* const sfmisc = sh.sf_misc[SF ? wib : 0];
    auto sfrow = [&](int stage, int r) -> float* { return sh.sf_rows[SF ? wib : 0][SF ? stage * kSfRows + r : 0]; };
    if constexpr (SF) {
        if (lane == 0) sh.sf_all[wib] = sf ? 1 : 0;
        if (sf) {
            // coefficients per send (lane 0: the direct send, lane 1: this slot's), histories per (send, channel)
            const oalsfx_source_params& SGp = ctx.source[inst];
            const oalsfx_source_state& SGs = ctx.source_state[inst];
            if (lane < 2) {
                const oalsfx_send_params& sp = lane == 0 ? SGp.direct : SGp.aux[slot];
                float* t = sfmisc + sfm::TAB + 12 * lane;
                t[0] = sp.lp.b0; t[1] = sp.lp.b1; t[2] = sp.lp.b2; t[3] = sp.hp.b0; t[4] = sp.hp.b1; t[5] = sp.hp.b2;
                t[6] = sp.lp.a1; t[7] = sp.lp.a2; t[8] = sp.hp.a1; t[9] = sp.hp.a2;
                const bool enabled = lane == 0 || sp.out_channels != 0;
                reinterpret_cast<int*>(t)[10] = enabled ? (((sp.filter_type & OALSFX_AF_LOW_PASS) ? 1 : 0) | ((sp.filter_type & OALSFX_AF_HIGH_PASS) ? 2 : 0) | 4) : 0;
            }
            if (lane < kSfRows) {
                const int sd = lane / CH, c = lane % CH, at = sd ? 1 + slot : 0;
                const oalsfx_hist_t lp = SGs.lp[at][c], hp = SGs.hp[at][c];
                float* h = sfmisc + sfm::HIST + 6 * lane; // x: input, y: output of the first shelf = input of the second, z: output of the second
                h[0] = lp.x[0]; h[1] = lp.x[1]; h[2] = lp.y[0]; h[3] = lp.y[1]; h[4] = hp.y[0]; h[5] = hp.y[1];
            }
        }
    }
    // after the send-filter pre-pass an instance with a filter reads its sends' planes, any other instance the raw input
    const bool filtered = !sf && (flags & kFiltered) != 0 && has_filter;
    const float* src = ctx.raw_src + static_cast<size_t>(inst) * ctx.io_stride;
    const float* wsrc = src;
    if (filtered) {
        src = ctx.src + static_cast<size_t>(inst) * ctx.src_stride;
        wsrc = ctx.wet_src + static_cast<size_t>(inst) * ctx.src_stride;
    }
    float* dst = ctx.dst + static_cast<size_t>(inst) * ctx.io_stride;
    float* mixbuf = ctx.mixbuf ? ctx.mixbuf + static_cast<size_t>(inst) * nch * OALSFX_MAX_CHUNK : nullptr;
    const float b2a = 0.288675134595F;

    // Software pipeline: inputs of tile k+1 are requested before tile k is computed (every tap is >= 2 tiles away).
    v4f n_e = {0, 0, 0, 0}, n_a = n_e, n_el = n_e, n_lt = n_e, n_la = n_e, n_ll = n_e;
    float n_in0 = 0.0F, n_in1 = 0.0F;
    float n_inv[MC ? 8 : 1] = {}; // multichannel: the frame's input channels
    float n_w0 = 0.0F, n_w1 = 0.0F;
    float n_wv[MC ? 8 : 1] = {};
    // XF: which tap table is current in a tile: the one being faded in once the fade is through
    auto tapbase = [&](int tile) -> int { return (XF && fc0 < OALSFX_RV_FADE_SAMPLES && fc0 + (tile << 6) >= OALSFX_RV_FADE_SAMPLES) ? kTapN : static_cast<int>(ut::TAP4); };
    auto load4 = [&](unsigned t4x, int group, auto ring, int base = ut::TAP4) -> v4f {
        constexpr int r = decltype(ring)::value;
        const v4u d = *reinterpret_cast<const v4u*>(utu + base + 4 * group);
        const v4u lo = *reinterpret_cast<const v4u*>(utu + ut::LO + 4 * r);
        const unsigned bm = utu[ut::BMASK + r];
        v4f v;
        v.x = ld_ring<r>(slab_b, ((t4x - d.x) & bm) | lo.x);
        v.y = ld_ring<r>(slab_b, ((t4x - d.y) & bm) | lo.y);
        v.z = ld_ring<r>(slab_b, ((t4x - d.z) & bm) | lo.z);
        v.w = ld_ring<r>(slab_b, ((t4x - d.w) & bm) | lo.w);
        return v;
    };
    // AW: the aligned windows of the long rings' four tap groups (early taps, early line, late taps, late line)
    constexpr bool AW = OALSFX_AW && FP && !HY && !MD && !ST && !MC && !XF; // (in the multichannel plain build, measured: the sixteen registers spill, 65 -> 73 us for quad)
    v4f w_e = {0, 0, 0, 0}, w_el = w_e, w_lt = w_e, w_ll = w_e; // per group: the window before the one requested last
    // the window [A + 256, A + 512) for the tile whose first sample stands at byte position tile4, A = (tile4 - tap) rounded down to 256
    auto load4w = [&](unsigned tile4, int group, auto ring) -> v4f {
        constexpr int r = decltype(ring)::value;
        const v4u d = *reinterpret_cast<const v4u*>(utu + ut::TAP4 + 4 * group);
        const v4u lo = *reinterpret_cast<const v4u*>(utu + ut::LO + 4 * r);
        const unsigned bm = utu[ut::BMASK + r];
        const unsigned l4 = 256u + 4u * static_cast<unsigned>(lane);
        v4f v;
        v.x = ld_ring<r>(slab_b, ((((tile4 - d.x) & ~255u) + l4) & bm) | lo.x);
        v.y = ld_ring<r>(slab_b, ((((tile4 - d.y) & ~255u) + l4) & bm) | lo.y);
        v.z = ld_ring<r>(slab_b, ((((tile4 - d.z) & ~255u) + l4) & bm) | lo.z);
        v.w = ld_ring<r>(slab_b, ((((tile4 - d.w) & ~255u) + l4) & bm) | lo.w);
        return v;
    };
    // lane L's sample stands k + L samples into the two windows, k = (tile4 - tap) % 256 / 4
    auto rotate4 = [&](const v4f& before, const v4f& last, unsigned tile4, int group) -> v4f {
        const v4u d = *reinterpret_cast<const v4u*>(utu + ut::TAP4 + 4 * group);
        auto one = [&](float b, float l, unsigned dj) -> float {
            // (lane j holds entry j of either window; entries k.. of the one before and ..k-1 of the last are wanted: merged per lane first,
            // then one rotation by k lanes)
            const int k = static_cast<int>(((tile4 - dj) & 255u) >> 2);
            const float merged = lane >= k ? b : l;
            return __uint_as_float(static_cast<unsigned>(__builtin_amdgcn_ds_bpermute(((lane + k) & 63) << 2, static_cast<int>(__float_as_uint(merged)))));
        };
        v4f v;
        v.x = one(before.x, last.x, d.x); v.y = one(before.y, last.y, d.y); v.z = one(before.z, last.z, d.z); v.w = one(before.w, last.w, d.w);
        return v;
    };
    // modulated late line (reference calc_modulation_delays, src/oalsfxpp.cpp:7443-7470): delay of this lane's sample in
    // the tile after the ones already prepared; the smoother's chain is strictly sequential, tile after tile
    // ST build: 8 hand-over rows behind the table; the modulation smoother's row (MD) lives only inside next_mod_delays and shares the first
    auto strow = [&](int k) -> float* { return utf + ut::SIZE + k * kRow; };
    float* modrow = utf + ut::SIZE;
    int mod_pos = v_modidx; // the modulator's index at the first sample of the next tile to prepare (the state keeps it in [0, range))
    auto next_mod_delays = [&](int samples_or_64) -> int { // RG: what the tile holds (a ragged call's last tile holds fewer than 64)
        const int samples = RG ? samples_or_64 : 64;
        if (lane == 0) {
            float r = mod_f;
            const float depth = mod_depth, coeff = mod_coeff;
            for (int i = 0; i < samples; ++i) {
                r = lerpf(r, depth, coeff);
                modrow[i] = r;
            }
        }
        wave_sync();
        const float fv = modrow[lane];
        mod_f = modrow[samples - 1];
        wave_sync();
        // (the index of this lane's sample modulo the range: carried from tile to tile by additions where the range is at least a tile
        // long -- the division by a run-time divisor is some thirty-five vector instructions)
        int index;
        if (!RG && v_modrange >= 64) { // (the ragged builds: the branch costs the stereo one two registers it does not have)
            index = mod_pos + lane;
            if (index >= v_modrange) index -= v_modrange;
            mod_pos += 64;
            if (mod_pos >= v_modrange) mod_pos -= v_modrange;
        } else {
            index = (mod_pos + lane) % v_modrange;
            mod_pos += 64;
        }
        const float sinus = glibc_sinf(6.28318530717958647692F * index / v_modrange);
        return lround_away(fv * sinus);
    };
    int md_next = 0, md_cur = 0;
    // the frame of a tile (its input half runs one iteration before its late half)
    auto issue_input = [&](int posx) {
        const int px = min(posx, frames - 1);
        if (MC) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < nch) n_inv[MC ? c : 0] = src[static_cast<size_t>(px) * nch + c];
            if (filtered) {
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    if (c < nch) n_wv[MC ? c : 0] = wsrc[static_cast<size_t>(px) * nch + c];
            }
        } else if (CH == 2) {
            if (OALSFX_NT & 32) {
                const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(src + static_cast<size_t>(px) * 2));
                n_in0 = v.x; n_in1 = v.y;
            } else {
                const float2 v = *reinterpret_cast<const float2*>(src + static_cast<size_t>(px) * 2);
                n_in0 = v.x; n_in1 = v.y;
            }
            if (filtered) {
                const float2 u = *reinterpret_cast<const float2*>(wsrc + static_cast<size_t>(px) * 2);
                n_w0 = u.x; n_w1 = u.y;
            }
        } else {
            n_in0 = src[px];
            if (filtered) n_w0 = wsrc[px];
        }
    };
    // the ring requests of a tile in three parts (S1, S3, S5 of the iteration before its late half)
    auto issue_taps_a = [&](unsigned t4x, int base) {
        if (AW) n_e = load4w(t4x - 4u * static_cast<unsigned>(lane), 0, RingId<OALSFX_RV_MAIN>{});
        else if (!HY || !(late_mask & 1u)) n_e = load4(t4x, 0, RingId<OALSFX_RV_MAIN>{}, base);
        if (!HY || !(late_mask & 2u)) n_a = load4(t4x, 1, RingId<OALSFX_RV_EARLY_AP>{}, base);
    };
    auto issue_taps_b = [&](unsigned t4x, int base) {
        if (AW) {
            const unsigned tile4 = t4x - 4u * static_cast<unsigned>(lane);
            n_el = load4w(tile4, 2, RingId<OALSFX_RV_EARLY_LINE>{});
            n_lt = load4w(tile4, 3, RingId<OALSFX_RV_MAIN>{});
            n_ll = load4w(tile4, 5, RingId<OALSFX_RV_LATE_LINE>{});
            return;
        }
        if (!HY || !(late_mask & 4u)) n_el = load4(t4x, 2, RingId<OALSFX_RV_EARLY_LINE>{}, base);
        if (!HY || !(late_mask & 8u)) n_lt = load4(t4x, 3, RingId<OALSFX_RV_MAIN>{}, base);
        if (!HY || !(late_mask & 32u)) n_ll = load4(MD ? t4x - 4u * static_cast<unsigned>(md_next) : t4x, 5, RingId<OALSFX_RV_LATE_LINE>{}, base);
    };
    auto issue_taps_c = [&](unsigned t4x, int base) {
        if (!HY || !(late_mask & 16u)) n_la = load4(t4x, 4, RingId<OALSFX_RV_LATE_AP>{}, base);
    };
    auto store4 = [&](unsigned p4, auto ring, float v0, float v1, float v2, float v3) {
        constexpr int r = decltype(ring)::value;
        const v4u lo = *reinterpret_cast<const v4u*>(utu + ut::LO + 4 * r);
#ifdef OALSFX_ABLATE_STORE_ALIGN // timing experiment (scripts/ablate_store_align.sh): ring stores rounded down to whole lines, results wrong
        const unsigned wp = ((((p4 - 4u * static_cast<unsigned>(lane)) & ~static_cast<unsigned>(OALSFX_ABLATE_STORE_ALIGN - 1)) + 4u * static_cast<unsigned>(lane))) & utu[ut::BMASK + r];
#else
        const unsigned wp = p4 & utu[ut::BMASK + r];
#endif
        st_ring<r>(slab_b, wp | lo.x, v0); st_ring<r>(slab_b, wp | lo.y, v1); st_ring<r>(slab_b, wp | lo.z, v2); st_ring<r>(slab_b, wp | lo.w, v3);
    };
    // CR: a store site's tile as whole lines (see the template parameter).  `rot`: lane L holds the value of the sample r places before
    // its own (lanes below r: of the tile's last r samples); `before`: what those lanes held a tile ago.
    static_assert(CR == 0 || (AW && !SF), "line-aligned stores: the plain FP builds");
    auto cr_lane_before = [&](unsigned r) -> int { return ((lane - static_cast<int>(r)) & 63) << 2; };
    auto cr_rot = [&](int from4, float v) -> float { return __uint_as_float(static_cast<unsigned>(__builtin_amdgcn_ds_bpermute(from4, static_cast<int>(__float_as_uint(v))))); };
    // (L: the samples the tile holds -- 64, or fewer in a ragged call's last tile, whose lanes from r + L on hold nothing to store and
    // whose own last samples, if they reach past the window, are the lanes below r + L - 64 of `rot`)
    auto carry_store = [&](unsigned tile4, auto ring, unsigned r, const v4f& rot, const v4f& before, bool head, bool tail, int L) {
        constexpr int rg = decltype(ring)::value;
        const v4u lo = *reinterpret_cast<const v4u*>(utu + ut::LO + 4 * rg);
        const unsigned bm = utu[ut::BMASK + rg];
        const bool held = lane < static_cast<int>(r);
        const int rl = static_cast<int>(r) + (RG ? L : 64);
        const unsigned at = tile4 - 4u * r + 4u * static_cast<unsigned>(lane);
#ifdef OALSFX_CR_ABLATE // timing experiment: no partial lines at the call's ends either (results wrong)
        head = tail = false;
#endif
        if (!(head && held) && (!RG || lane < rl)) { // (the call's first tile: the call before has written what lies in front of it)
            const unsigned wp = at & bm;
            st_ring<rg>(slab_b, wp | lo.x, held ? before.x : rot.x); st_ring<rg>(slab_b, wp | lo.y, held ? before.y : rot.y);
            st_ring<rg>(slab_b, wp | lo.z, held ? before.z : rot.z); st_ring<rg>(slab_b, wp | lo.w, held ? before.w : rot.w);
        }
        if (tail && (RG ? lane < rl - 64 : held)) { // the call's last tile: nothing comes behind it to take its last samples along
            const unsigned wp = (at + 256u) & bm;
            st_ring<rg>(slab_b, wp | lo.x, rot.x); st_ring<rg>(slab_b, wp | lo.y, rot.y); st_ring<rg>(slab_b, wp | lo.z, rot.z); st_ring<rg>(slab_b, wp | lo.w, rot.w);
        }
    };
    // ... held back in LDS (CR == 2): site 0 main delay, 1 early all-pass, 2 early line, 3 late all-pass, 4 late line.  Lane L < r reads
    // and writes entry L of its values' rows, nobody else's: no hand-over between lanes.
    auto cr_tail = [&](int site) -> v4f* { return reinterpret_cast<v4f*>(&sh.cr_tails[CR == 2 ? wib : 0][CR == 2 ? site : 0][lane & 31][0]); }; // (one 16-byte record per lane and site)
    auto carry_store_lds = [&](unsigned tile4, auto ring, int site, const v4f& rot, bool head, bool tail, int L) {
        v4f* t = cr_tail(site);
        const unsigned r = utu[ut::CR_R];
        const v4f before = *t;
        carry_store(tile4, ring, r, rot, before, head, tail, L);
        if (lane < static_cast<int>(r)) *t = rot;
    };
    v4f cr_feed = {0, 0, 0, 0}; // CR >= 1: the late feed's values of the tile before, rotated (lanes below r: its last r samples)
    // chain lanes of the chain phases: lane -> (wave cw, line cc) for lane < 4 * NW
    const int cw = (lane >> 2) & (NW - 1), cc = lane & 3;
    float* crowI0 = lds_all[cw] + (0 * 4 + cc) * kRow;
    float* crowI1 = lds_all[cw] + (1 * 4 + cc) * kRow;
    float* crowI2 = lds_all[cw] + (2 * 4 + cc) * kRow;
    float* crowL1 = lds_all[cw] + (4 * 4 + cc) * kRow;
    float* crowL2 = lds_all[cw] + (5 * 4 + cc) * kRow;
    float* cdat = chain_all[cw][cc];

    if (CR != 0 && go) {
        if (lane == 0) {
            utu[ut::CR_R] = static_cast<unsigned>(offset) & 31u;
            utu[ut::CR_RF] = (static_cast<unsigned>(offset) - (utu[ut::FEED4] >> 2)) & 31u;
        }
        wave_sync();
    }
    stamp(); // [2] tables written
    // the first tile's inputs are requested before the workgroup barrier: they travel while the shelves of tile 0 run
    if (go) {
        if (FP && !(flags & kFiltered)) { n_in0 = early_in0; n_in1 = early_in1; }
        else issue_input(lane);
        if (AW) w_e = load4w((static_cast<unsigned>(offset) << 2) - 256u, 0, RingId<OALSFX_RV_MAIN>{}); // the window before the first tile's
        issue_taps_a(static_cast<unsigned>(offset + lane) << 2, tapbase(0));
    }
    stamp(); // [3] first requests issued
    lds_barrier(); // tables, chain data and go flags are in place
    stamp(); // [4]
    bool any_go = false;
    bool any_eax = false; // the second-shelf phases (and their barriers) exist when some instance of the group needs them
    bool any_sf = false;  // SF: some instance of the group filters its sends in here: the loop starts one iteration earlier
#pragma unroll
    for (int k = 0; k < NW; ++k) { any_go |= go_all[k] != 0; any_eax |= eax_all[k] != 0; if (SF) any_sf |= sh.sf_all[k] != 0; }
    const bool chain_on = (lane < 4 * NW) && go_all[cw] != 0;
    const bool chain2_on = chain_on && eax_all[cw] != 0;
    // which wavefront runs chain phase p: rotated per workgroup so that the co-resident workgroups of a CU do not all
    // put the same phase on the same SIMD
    const int duty = (wib - group) & (NW - 1);

    const int tiles = any_go ? (RG ? (frames + 63) >> 6 : frames >> 6) : 0; // a workgroup without a steady instance skips the cooperative loop altogether
    // The tile loop is skewed by one tile: iteration `it` runs the input half of tile it (P1, C1, P2, C2: send mix and input shelves, which
    // depend on nothing but the input frames) together with the late half of tile it - 1 (P3, C3, P4, C4, P5: everything that touches the
    // delay lines).  The chain phases of the two halves run at the same time on two different wavefronts of the workgroup, and a tile
    // costs four workgroup barriers and two chain-phase latencies instead of eight and four; the first tile's shelves run while its
    // ring requests travel.  The shelves' output of tile it waits in its own rows (rowI) for the next iteration's P3.
    float o0 = 0.0F, o1 = 0.0F;
    float outv[MC ? 8 : 1] = {}; // multichannel: the output frame being accumulated (of the tile whose late half runs)
    // XF: the blocks of an instance in transition (reference ReverbEffectState::do_process, src/oalsfxpp.cpp:6088-6096): at most 256 frames,
    // cut where a cross-fade ends; every block takes a new step for each output gain (MixHelpers::mix, src/oalsfxpp.cpp:2752-2798)
    int fc = fc0, blk_start = 0, blk_end = 0, blk_counter = 1;
    float g_step = 0.0F, g_run = 0.0F;
    bool g_ramp = false;
    unsigned long long ramp_mask = 0ULL;
    // ... and where the stepped gains of a tile go: rows of the late half that are idle in S5, and hand-over rows
    auto grow = [&](int q) -> float* { return q < 12 ? rowL(q >> 2, q & 3) + 4 : utf + ut::SIZE + (2 + q - 12) * kRow + 4; };
    // What the call leaves for the next one: the canonical state and, from an FP build, the hot record.  In an FP build this runs in the
    // call's last iteration, in front of S5 (OALSFX_EARLY_HANDBACK): every filter history is final once that iteration's chain phases are
    // through -- S5 only writes delay lines and output frames --, so the stores travel while S5 computes instead of standing between the
    // last ring store and the word that hands the instance on.
    constexpr bool EH = OALSFX_EARLY_HANDBACK && FP && !SF && !ST; // (the send-filter and short-tap builds have no registers for it: they spill)
    auto hand_back = [&]() {
        if (go) {
            if (lane < 4) {
                const float* ch = chain_all[wib][lane];
                S.lp[lane].x[0] = ch[coop::LPX0]; S.lp[lane].x[1] = ch[coop::LPX1];
                S.lp[lane].y[0] = ch[coop::LPY0]; S.lp[lane].y[1] = ch[coop::LPY1];
                if (eax) {
                    S.hp[lane].x[0] = ch[coop::LPY0]; S.hp[lane].x[1] = ch[coop::LPY1];
                    S.hp[lane].y[0] = ch[coop::HPY0]; S.hp[lane].y[1] = ch[coop::HPY1];
                }
                S.t60[lane][0][0] = ch[coop::T60X]; S.t60[lane][0][1] = ch[coop::T60O1];
                S.t60[lane][1][0] = ch[coop::T60O1]; S.t60[lane][1][1] = ch[coop::T60O2];
            }
            if (lane == 0) {
                S.mod_index = static_cast<int>((static_cast<long long>(v_modidx) + frames) % v_modrange);
                S.offset = offset + frames;
                if (MD && mod_on) S.mod_filter = mod_f;
            }
            if (XF && xf_active) {
                // the transition's own state: gains where the ramps left them, the fade count, the taps once they are faded in, the change
                // marked as seen; and whether the instance ends the call settled and at rest
                if (q_valid) {
                    if (q_stage) S.late_cur_gain[q_line][q_chan] = g_cur;
                    else S.early_cur_gain[q_line][q_chan] = g_cur;
                }
                const bool fade_over = fc >= OALSFX_RV_FADE_SAMPLES;
                if (fc0 < OALSFX_RV_FADE_SAMPLES && fade_over && lane < 24) (&S.cur_early_tap[0])[lane] = static_cast<int32_t>(utu[kTapN + lane] >> 2);
                if (lane == 0) {
                    S.fade_count = fc;
                    S.mod_range = v_modrange;
                    SS.seen_seq = SP.update_seq;
                }
                if (ctx.exact) {
                    const unsigned level = gains_rest_level(q_valid, g_cur, g_tgt);
                    if (lane == 0) ctx.exact[sidx] = fade_over ? level : 0u;
                }
            }
            if (SF && sf) {
                // the send filters' histories as the recurrence lanes left them (the second shelf's input history is the first one's output)
                if (lane < kSfRows) {
                    const int sd = lane / CH, c = lane % CH, at = sd ? 1 + slot : 0;
                    if (reinterpret_cast<const int*>(sfmisc + sfm::TAB + 12 * sd)[10] & 4) {
                        const float* h = sfmisc + sfm::HIST + 6 * lane;
                        oalsfx_source_state& SGs = ctx.source_state[inst];
                        oalsfx_hist_t lp, hp;
                        lp.x[0] = h[0]; lp.x[1] = h[1]; lp.y[0] = h[2]; lp.y[1] = h[3];
                        hp.x[0] = h[2]; hp.x[1] = h[3]; hp.y[0] = h[4]; hp.y[1] = h[5];
                        SGs.lp[at][c] = lp;
                        SGs.hp[at][c] = hp;
                    }
                }
            } else if (FP && !RG) {
                if (first && !filtered && lane < nch)
                    send_history_follow_values(ctx, inst, lane, send_mask, lane == 0 ? hist_new[0] : hist_new[CH - 1], lane == 0 ? hist_old[0] : hist_old[CH - 1]);
            } else if (first && !filtered && lane < nch) send_history_follow(ctx, inst, lane, nch, frames, src);
        }
        if constexpr (FP) {
            // ---- the hot record for the next call: histories and stamp always, the tables when they were rebuilt ----
            if (go) {
                if (lane == 0) {
                    miscu[hot::M_EPOCH] = epoch_now;
                    miscu[hot::M_OFFSET] = static_cast<unsigned>(offset + frames);
                    miscu[hot::M_MOD_F] = __float_as_uint(mod_f);
                    miscu[hot::M_MOD_INDEX] = static_cast<unsigned>((static_cast<long long>(v_modidx) + frames) % v_modrange);
                    miscu[hot::M_MOD_ON] = (MD && ((mod_depth != 0.0F) || (mod_f != 0.0F))) ? 1u : 0u;
                }
                wave_sync();
                v4u* rec = reinterpret_cast<v4u*>(ctx.hot + sidx * hot::SIZE);
                if (lane >= 48) rec[lane] = *reinterpret_cast<const v4u*>(miscu + 4 * (lane - 48));
                else if (lane >= 32) rec[lane] = *reinterpret_cast<const v4u*>(&chain_all[wib][0][0] + 4 * (lane - 32));
                else if (!hit) rec[lane] = *reinterpret_cast<const v4u*>(utu + 4 * lane);
            }
        }
    };
    for (int it = (SF && any_sf) ? -1 : 0; it <= tiles && tiles > 0; ++it) {
        const int ta = it, tb = it - 1, tc = it + 1; // tc (SF): the tile whose frame goes through the send filters in this iteration
        const bool has_a = ta >= 0 && ta < tiles, has_b = tb >= 0, has_c = SF && tc < tiles;
        const int pos_a = (ta << 6) + lane, pos_b = (tb << 6) + lane;
        // RG, the build for calls that are not a whole number of tiles: the last tile holds fewer samples; its lanes from L on
        // compute along but store nothing, and the recurrences stop at L.  (Its own build: with L a variable the chain loops and
        // the predicated stores cost the whole-tile case 6 %.)
        const int La = RG ? min(64, frames - (ta << 6)) : 64;
        const int Lb = RG ? min(64, frames - (tb << 6)) : 64;
        const bool act = RG ? lane < Lb : true;
        const unsigned t4 = static_cast<unsigned>(offset + pos_b) << 2; // the late half's tile
        const unsigned tile_b4 = static_cast<unsigned>(offset + (tb << 6)) << 2; // ... its first sample (CR)
        const bool head_b = tb == 0, tail_b = tb == tiles - 1;
        float oa0 = 0.0F, oa1 = 0.0F;
        float outva[MC ? 8 : 1] = {}; // the dry mix (or the running mix of the slots before) of tile ta
        v4f p_e = n_e, p_a = n_a, p_el = n_el, p_lt = n_lt, p_la = n_la, p_ll = n_ll;
        if (AW && go && has_b) {
            // the long rings' samples of tile tb: each lane's from the window requested for the tile or from the one before it
            const unsigned tile4 = static_cast<unsigned>(offset + (tb << 6)) << 2;
            p_e = rotate4(w_e, n_e, tile4, 0); p_el = rotate4(w_el, n_el, tile4, 2); p_lt = rotate4(w_lt, n_lt, tile4, 3); p_ll = rotate4(w_ll, n_ll, tile4, 5);
            w_e = n_e; w_el = n_el; w_lt = n_lt; w_ll = n_ll;
        }
        const int xg = eax ? 0 : 2; // where the shelves left their output
        // XF: is the late half's tile one in which the taps are cross-faded?  mu: how far, per sample; q_*: the taps being faded in
        const bool xf_faded = XF && xf_active && has_b && fc0 + (tb << 6) < OALSFX_RV_FADE_SAMPLES;
        const float mu = static_cast<float>(fc0 + pos_b) * (1.0F / OALSFX_RV_FADE_SAMPLES);
        v4f q_e = {0, 0, 0, 0}, q_a = q_e, q_lt = q_e, q_el = q_e;
        auto mix4 = [&](v4f& a, const v4f& b) {
            a.x = lerpf(a.x, b.x, mu); a.y = lerpf(a.y, b.y, mu); a.z = lerpf(a.z, b.z, mu); a.w = lerpf(a.w, b.w, mu);
        };
        const int tb_base = tapbase(tb); // the tap table that is current in the late half's tile

        if (go) {
            if (MD) {
                md_cur = md_next; // the delays of tile tb
                if (mod_on && has_a) md_next = next_mod_delays(min(64, frames - (ta << 6))); // ... of tile ta, whose late-line requests go out below
            }
            if (HY && has_b) {
                // groups with a tap closer than two tiles: requested now, after the previous tile's stores
                const int base = tapbase(tb);
                if (late_mask & 1u) p_e = load4(t4, 0, RingId<OALSFX_RV_MAIN>{}, base);
                if (late_mask & 2u) p_a = load4(t4, 1, RingId<OALSFX_RV_EARLY_AP>{}, base);
                if (late_mask & 4u) p_el = load4(t4, 2, RingId<OALSFX_RV_EARLY_LINE>{}, base);
                if (late_mask & 8u) p_lt = load4(t4, 3, RingId<OALSFX_RV_MAIN>{}, base);
                if (late_mask & 16u) p_la = load4(t4, 4, RingId<OALSFX_RV_LATE_AP>{}, base);
                if (late_mask & 32u) p_ll = load4(MD ? t4 - 4u * static_cast<unsigned>(md_cur) : t4, 5, RingId<OALSFX_RV_LATE_LINE>{}, base);
            }
            if (XF && xf_faded) {
                // a cross-fading tile (reference delay_out_faded, src/oalsfxpp.cpp:7378-7399): the taps being faded in are requested here,
                // behind the previous tile's stores, and mixed with the current ones sample by sample: the late line at once (its taps
                // are a tile away), the others where they are used, behind the hand-over inside the tile that either set may need
                q_e = load4(t4, 0, RingId<OALSFX_RV_MAIN>{}, kTapN); q_a = load4(t4, 1, RingId<OALSFX_RV_EARLY_AP>{}, kTapN);
                q_lt = load4(t4, 3, RingId<OALSFX_RV_MAIN>{}, kTapN);
                q_el = load4(t4, 2, RingId<OALSFX_RV_EARLY_LINE>{}, kTapN);
                const v4f q_ll = load4(t4 - 4u * static_cast<unsigned>(md_cur), 5, RingId<OALSFX_RV_LATE_LINE>{}, kTapN);
                mix4(p_ll, q_ll);
            }
            // the ring requests of tile ta (its late half runs in the next iteration) go out in three parts (top and end of S1, S3): a wavefront
            // that issues all 24 in one go sits in the issue queue while its siblings and its own arithmetic wait
            if (has_a && it > 0) issue_taps_a(t4 + 256u, tapbase(ta)); // (the prologue issued tile 0's)
            __builtin_amdgcn_sched_barrier(0); // keep the requests up here: the scheduler would sink them next to their first use
        }
        // ---------------- S1, late half: P3(tb): main delay write, early reflections, late taps, T60 first feed-forward ----------------
        v2f e01 = {0, 0}, e23 = {0, 0};
        float dg = 0.0F, ac = 0.0F, sx = 0.0F, sy = 0.0F;
        if (go && has_b) {
            if constexpr (CR == 2) {
                // (the shelves' output lies in LDS rows: read r places back, no lane rotation)
                const int from = 4 + (cr_lane_before(utu[ut::CR_R]) >> 2);
                carry_store_lds(tile_b4, RingId<OALSFX_RV_MAIN>{}, 0, v4f{rowI(xg, 0)[from], rowI(xg, 1)[from], rowI(xg, 2)[from], rowI(xg, 3)[from]}, head_b, tail_b, Lb);
            } else if (act) store4(t4, RingId<OALSFX_RV_MAIN>{}, rowI(xg, 0)[4 + lane], rowI(xg, 1)[4 + lane], rowI(xg, 2)[4 + lane], rowI(xg, 3)[4 + lane]);
            wave_sync();
            const v4f misc = *reinterpret_cast<const v4f*>(utf + ut::MISC);
            dg = misc.x; ac = misc.y; sx = misc.z; sy = misc.w;
            const v4f ec = *reinterpret_cast<const v4f*>(utf + ut::ECOEF);
            const v4f elc = *reinterpret_cast<const v4f*>(utf + ut::ELCOEF);
            if (ST && (short_mask & 1u)) {
                // early taps shorter than the tile read what an earlier lane just wrote to the main delay: the shelves' output
                auto hand_over = [&](v4f& v, int base) {
                    const v4u d = *reinterpret_cast<const v4u*>(utu + base);
                    const int e0 = static_cast<int>(d.x >> 2), e1 = static_cast<int>(d.y >> 2), e2 = static_cast<int>(d.z >> 2), e3 = static_cast<int>(d.w >> 2);
                    if (lane >= e0) v.x = rowI(xg, 0)[4 + lane - e0];
                    if (lane >= e1) v.y = rowI(xg, 1)[4 + lane - e1];
                    if (lane >= e2) v.z = rowI(xg, 2)[4 + lane - e2];
                    if (lane >= e3) v.w = rowI(xg, 3)[4 + lane - e3];
                };
                hand_over(p_e, XF ? tb_base : static_cast<int>(ut::TAP4));
                if (XF && xf_faded) hand_over(q_e, kTapN);
            }
            if (XF && xf_faded) mix4(p_e, q_e);
            if (ST && (short_mask & 2u)) {
                // all-pass offsets shorter than the tile: the lanes whose sources lie in earlier tiles are right from the
                // start; every evaluation ahead of the real one makes the next `shortest offset` lanes right, which then hand
                // their all-pass ring values to the lanes that read them
                const v4u d = *reinterpret_cast<const v4u*>(utu + (XF ? tb_base : static_cast<int>(ut::TAP4)) + 4);
                const int o0x = static_cast<int>(d.x >> 2), o1x = static_cast<int>(d.y >> 2), o2x = static_cast<int>(d.z >> 2), o3x = static_cast<int>(d.w >> 2);
                int shortest_off = min(min(o0x, o1x), min(o2x, o3x));
                // XF, a cross-fading tile: the set being faded in hands over by its own offsets
                const v4u dn = *reinterpret_cast<const v4u*>(utu + (XF ? kTapN : static_cast<int>(ut::TAP4)) + 4);
                const int n0x = static_cast<int>(dn.x >> 2), n1x = static_cast<int>(dn.y >> 2), n2x = static_cast<int>(dn.z >> 2), n3x = static_cast<int>(dn.w >> 2);
                if (XF && xf_faded) shortest_off = min(shortest_off, min(min(n0x, n1x), min(n2x, n3x)));
                const int ahead = 63 / shortest_off;
                const v2f pf01 = v2f{p_e.x, p_e.y} * v2f{ec.x, ec.y};
                const v2f pf23 = v2f{p_e.z, p_e.w} * v2f{ec.z, ec.w};
                for (int k = 0; k < ahead; ++k) {
                    v4f a_now = p_a;
                    if (XF && xf_faded) mix4(a_now, q_a);
                    const v2f pv01 = v2f{a_now.x, a_now.y} - (ac * pf01);
                    const v2f pv23 = v2f{a_now.z, a_now.w} - (ac * pf23);
                    v2f pg01 = pf01 + (ac * pv01);
                    v2f pg23 = pf23 + (ac * pv23);
                    scatter2(pg01, pg23, sx, sy);
                    strow(0)[4 + lane] = pg01.x; strow(1)[4 + lane] = pg01.y; strow(2)[4 + lane] = pg23.x; strow(3)[4 + lane] = pg23.y;
                    wave_sync();
                    if (lane >= o0x) p_a.x = strow(0)[4 + lane - o0x];
                    if (lane >= o1x) p_a.y = strow(1)[4 + lane - o1x];
                    if (lane >= o2x) p_a.z = strow(2)[4 + lane - o2x];
                    if (lane >= o3x) p_a.w = strow(3)[4 + lane - o3x];
                    if (XF && xf_faded) {
                        if (lane >= n0x) q_a.x = strow(0)[4 + lane - n0x];
                        if (lane >= n1x) q_a.y = strow(1)[4 + lane - n1x];
                        if (lane >= n2x) q_a.z = strow(2)[4 + lane - n2x];
                        if (lane >= n3x) q_a.w = strow(3)[4 + lane - n3x];
                    }
                    wave_sync();
                }
            }
            if (XF && xf_faded) mix4(p_a, q_a);
            const v2f f01 = v2f{p_e.x, p_e.y} * v2f{ec.x, ec.y};
            const v2f f23 = v2f{p_e.z, p_e.w} * v2f{ec.z, ec.w};
            const v2f v01 = v2f{p_a.x, p_a.y} - (ac * f01);
            const v2f v23 = v2f{p_a.z, p_a.w} - (ac * f23);
            v2f g01 = f01 + (ac * v01);
            v2f g23 = f23 + (ac * v23);
            scatter2(g01, g23, sx, sy);
            if constexpr (CR == 2) {
                const int f4 = cr_lane_before(utu[ut::CR_R]);
                carry_store_lds(tile_b4, RingId<OALSFX_RV_EARLY_AP>{}, 1, v4f{cr_rot(f4, g01.x), cr_rot(f4, g01.y), cr_rot(f4, g23.x), cr_rot(f4, g23.y)}, head_b, tail_b, Lb);
                carry_store_lds(tile_b4, RingId<OALSFX_RV_EARLY_LINE>{}, 2, v4f{cr_rot(f4, v23.y), cr_rot(f4, v23.x), cr_rot(f4, v01.y), cr_rot(f4, v01.x)}, head_b, tail_b, Lb);
            } else if (act) {
                store4(t4, RingId<OALSFX_RV_EARLY_AP>{}, g01.x, g01.y, g23.x, g23.y);
                store4(t4, RingId<OALSFX_RV_EARLY_LINE>{}, v23.y, v23.x, v01.y, v01.x);
            }
            if (ST && (short_mask & 4u)) {
                // early-line offsets shorter than the tile (sampling rates below 16 kHz) read what an earlier lane just wrote to the line: the
                // reversed all-pass output, which does not depend on the line (no loop to settle: one hand-over; an offset of zero reads the
                // lane's own, as the reference's just-written slot does)
                strow(0)[4 + lane] = v23.y; strow(1)[4 + lane] = v23.x; strow(2)[4 + lane] = v01.y; strow(3)[4 + lane] = v01.x;
                wave_sync();
                auto hand_over = [&](v4f& v, int base) {
                    const v4u d = *reinterpret_cast<const v4u*>(utu + base + 8);
                    const int e0 = static_cast<int>(d.x >> 2), e1 = static_cast<int>(d.y >> 2), e2 = static_cast<int>(d.z >> 2), e3 = static_cast<int>(d.w >> 2);
                    if (lane >= e0) v.x = strow(0)[4 + lane - e0];
                    if (lane >= e1) v.y = strow(1)[4 + lane - e1];
                    if (lane >= e2) v.z = strow(2)[4 + lane - e2];
                    if (lane >= e3) v.w = strow(3)[4 + lane - e3];
                };
                hand_over(p_el, XF ? tb_base : static_cast<int>(ut::TAP4));
                if (XF && xf_faded) hand_over(q_el, kTapN);
                wave_sync();
            }
            if (XF && xf_faded) mix4(p_el, q_el);
            e01 = v01 + (v2f{p_el.x, p_el.y} * v2f{elc.x, elc.y});
            e23 = v23 + (v2f{p_el.z, p_el.w} * v2f{elc.z, elc.w});
            {
                v2f r01 = {e23.y, e23.x}, r23 = {e01.y, e01.x};
                scatter2(r01, r23, sx, sy);
                if constexpr (CR >= 1) {
                    const unsigned rf = utu[ut::CR_RF];
                    const int f4 = cr_lane_before(rf);
                    const v4f rot = {cr_rot(f4, r01.x), cr_rot(f4, r01.y), cr_rot(f4, r23.x), cr_rot(f4, r23.y)};
                    carry_store(tile_b4 - utu[ut::FEED4], RingId<OALSFX_RV_MAIN>{}, rf, rot, cr_feed, head_b, tail_b, Lb);
                    cr_feed = rot;
                } else if (act) store4(t4 - utu[ut::FEED4], RingId<OALSFX_RV_MAIN>{}, r01.x, r01.y, r23.x, r23.y);
                if (ST && (short_mask & 8u)) {
                    // late taps closer than a tile to the late feed read what an earlier lane just fed
                    strow(4)[4 + lane] = r01.x; strow(5)[4 + lane] = r01.y; strow(6)[4 + lane] = r23.x; strow(7)[4 + lane] = r23.y;
                    wave_sync();
                    auto hand_over = [&](v4f& v, int base) {
                        const v4u d = *reinterpret_cast<const v4u*>(utu + base + 12);
                        const unsigned f4 = utu[ut::FEED4];
                        const int l0 = static_cast<int>((d.x - f4) >> 2), l1 = static_cast<int>((d.y - f4) >> 2), l2 = static_cast<int>((d.z - f4) >> 2),
                                  l3 = static_cast<int>((d.w - f4) >> 2);
                        if (lane >= l0) v.x = strow(4)[4 + lane - l0];
                        if (lane >= l1) v.y = strow(5)[4 + lane - l1];
                        if (lane >= l2) v.z = strow(6)[4 + lane - l2];
                        if (lane >= l3) v.w = strow(7)[4 + lane - l3];
                    };
                    hand_over(p_lt, XF ? tb_base : static_cast<int>(ut::TAP4));
                    if (XF && xf_faded) hand_over(q_lt, kTapN);
                }
            }
            if (XF && xf_faded) mix4(p_lt, q_lt);
            const v2f u01 = (v2f{p_lt.x, p_lt.y} * dg) + v2f{p_ll.x, p_ll.y};
            const v2f u23 = (v2f{p_lt.z, p_lt.w} * dg) + v2f{p_ll.z, p_ll.w};
            if (lane < 4) rowL(0, lane)[3] = chain_all[wib][lane][coop::T60X];
            rowL(0, 0)[4 + lane] = u01.x; rowL(0, 1)[4 + lane] = u01.y; rowL(0, 2)[4 + lane] = u23.x; rowL(0, 3)[4 + lane] = u23.y;
            wave_sync();
            {
                const v4f c0 = *reinterpret_cast<const v4f*>(utf + ut::TL0);
                const v4f c1 = *reinterpret_cast<const v4f*>(utf + ut::TL1);
                const float* xa = rowL(0, 0) + 4 + lane; const float* xb = rowL(0, 1) + 4 + lane;
                const float* xc = rowL(0, 2) + 4 + lane; const float* xd = rowL(0, 3) + 4 + lane;
                const v2f w01 = (v2f{c0.x, c0.y} * v2f{xa[0], xb[0]}) + (v2f{c1.x, c1.y} * v2f{xa[-1], xb[-1]});
                const v2f w23 = (v2f{c0.z, c0.w} * v2f{xc[0], xd[0]}) + (v2f{c1.z, c1.w} * v2f{xc[-1], xd[-1]});
                rowL(1, 0)[4 + lane] = w01.x; rowL(1, 1)[4 + lane] = w01.y; rowL(1, 2)[4 + lane] = w23.x; rowL(1, 3)[4 + lane] = w23.y;
            }
            if (lane < 4) chain_all[wib][lane][coop::T60X] = rowL(0, lane)[4 + Lb - 1];
        }
        // ---------------- S1, input half: P1(ta): inputs, A-format, feed-forward half of the first shelf ----------------
        if (go && has_a) {
            float in[2] = {n_in0, n_in1};
            float win[2] = {filtered ? n_w0 : n_in0, filtered ? n_w1 : n_in1}; // what the auxiliary send sees
            if (SF && sf) {
                // send filters inside: the frame as the two sends see it, left in the second shelves' rows by the iteration before
#pragma unroll
                for (int c = 0; c < (MC ? 0 : CH); ++c) { in[c] = sfrow(1, c)[4 + lane]; win[c] = sfrow(1, CH + c)[4 + lane]; }
            }
            float inv[MC ? 8 : 1], winv[MC ? 8 : 1];
#pragma unroll
            for (int c = 0; c < (MC ? 8 : 1); ++c) { inv[c] = n_inv[c]; winv[c] = filtered ? n_wv[c] : n_inv[c]; }
            if (!(SF && sf) && ta + 1 < tiles) issue_input(pos_a + 64); // the next tile's frame, now that this one's is in `in`
            if (FP && !RG && !(SF && sf) && ta + 1 == tiles) {
                // the call's last two frames, for the histories of the pass-through send filters (no loads in the epilogue)
#pragma unroll
                for (int c = 0; c < (MC ? 0 : CH); ++c) {
                    hist_new[c] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(in[c]), 63));
                    hist_old[c] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(in[c]), 62));
                }
            }
            float wet[4] = {0.0F, 0.0F, 0.0F, 0.0F};
            if (MC) {
                // dry mix and B-format send, channel by channel (reference mix_source, src/oalsfxpp.cpp:2917-2982)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (c >= nch) continue;
                    if (first) {
#pragma unroll
                        for (int o = 0; o < 8; ++o)
                            if (aud_dir & (1ULL << (c * 8 + o))) outva[MC ? o : 0] += inv[MC ? c : 0] * utf[kMcBase + 64 + c * 8 + o];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (aud_aux & (1ULL << (c * 4 + k))) wet[k] += winv[MC ? c : 0] * utf[kMcBase + 128 + c * 4 + k];
                }
                if (!first) {
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (c < nch) outva[MC ? c : 0] = mixbuf[c * OALSFX_MAX_CHUNK + pos_a];
                }
            } else if (!first) {
                oa0 = mixbuf[pos_a];
                if (CH == 2) oa1 = mixbuf[OALSFX_MAX_CHUNK + pos_a];
            } else {
                const v4f gd = *reinterpret_cast<const v4f*>(utf + ut::GDIR);
                const float g[4] = {gd.x, gd.y, gd.z, gd.w};
#pragma unroll
                for (int c = 0; c < (MC ? 0 : CH); ++c) {
                    if (aud_dir & (1u << (c * 2 + 0))) oa0 += in[c] * g[c * 2 + 0];
                    if (CH == 2 && (aud_dir & (1u << (c * 2 + 1)))) oa1 += in[c] * g[c * 2 + 1];
                }
            }
#pragma unroll
            for (int c = 0; c < (MC ? 0 : CH); ++c) {
                const v4f ga = *reinterpret_cast<const v4f*>(utf + ut::GAUX + 4 * c);
                const float g[4] = {ga.x, ga.y, ga.z, ga.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (aud_aux & (1u << (c * 4 + k))) wet[k] += win[c] * g[k];
            }
            v2f a01 = {0.0F, 0.0F}, a23 = {0.0F, 0.0F};
            a01 = a01 + wet[0] * v2f{b2a, b2a}; a01 = a01 + wet[1] * v2f{b2a, -b2a}; a01 = a01 + wet[2] * v2f{b2a, -b2a}; a01 = a01 + wet[3] * v2f{b2a, b2a};
            a23 = a23 + wet[0] * v2f{b2a, b2a}; a23 = a23 + wet[1] * v2f{b2a, -b2a}; a23 = a23 + wet[2] * v2f{-b2a, b2a}; a23 = a23 + wet[3] * v2f{-b2a, -b2a};
            if (lane < 4) {
                const float* ch = chain_all[wib][lane];
                rowI(0, lane)[3] = ch[coop::LPX0]; rowI(0, lane)[2] = ch[coop::LPX1];
            }
            rowI(0, 0)[4 + lane] = a01.x; rowI(0, 1)[4 + lane] = a01.y; rowI(0, 2)[4 + lane] = a23.x; rowI(0, 3)[4 + lane] = a23.y;
            wave_sync();
            {
                const v4f bq = *reinterpret_cast<const v4f*>(utf + ut::LPB);
                const float* xa = rowI(0, 0) + 4 + lane; const float* xb = rowI(0, 1) + 4 + lane;
                const float* xc = rowI(0, 2) + 4 + lane; const float* xd = rowI(0, 3) + 4 + lane;
                const v2f u01 = ((bq.x * v2f{xa[0], xb[0]}) + (bq.y * v2f{xa[-1], xb[-1]})) + (bq.z * v2f{xa[-2], xb[-2]});
                const v2f u23 = ((bq.x * v2f{xc[0], xd[0]}) + (bq.y * v2f{xc[-1], xd[-1]})) + (bq.z * v2f{xc[-2], xd[-2]});
                rowI(1, 0)[4 + lane] = u01.x; rowI(1, 1)[4 + lane] = u01.y; rowI(1, 2)[4 + lane] = u23.x; rowI(1, 3)[4 + lane] = u23.y;
            }
            if (lane < 4) {
                float* ch = chain_all[wib][lane];
                ch[coop::LPX1] = rowI(0, lane)[4 + La - 2]; ch[coop::LPX0] = rowI(0, lane)[4 + La - 1]; // La == 1: [3] is the old newest sample
            }
        }
        // ---------------- S1, SF: the frame of tile tc through the first shelves' feed-forward halves (lane = frame) ----------------
        if (SF && sf && has_c) {
            const float x[2] = {n_in0, n_in1};
            if (tc + 1 < tiles) issue_input(((tc + 1) << 6) + lane);
#pragma unroll
            for (int c = 0; c < (MC ? 0 : CH); ++c) {
                // samples 0 and 1 need the send's own input history, which the recurrence lane holds: it writes them (and takes the tile's
                // last two inputs for the next tile's) from these edge samples
                if (lane < 2) sfmisc[sfm::EDGE + 4 * c + lane] = x[c];
                if (lane >= 62) sfmisc[sfm::EDGE + 4 * c + lane - 60] = x[c];
                const float xm1 = __shfl_up(x[c], 1), xm2 = __shfl_up(x[c], 2);
#pragma unroll
                for (int sd = 0; sd < 2; ++sd) {
                    const float* t = sfmisc + sfm::TAB + 12 * sd;
                    const int mode = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(t)[10]);
                    if (!(mode & 4)) continue;
                    float v = x[c];
                    if ((mode & 1) && lane >= 2) v = (t[0] * x[c]) + (t[1] * xm1) + (t[2] * xm2);
                    sfrow(0, sd * CH + c)[4 + lane] = v;
                }
            }
        }
        if (go && has_a) {
            // (all of a tile's requests at the top of S1, or the late all-pass group here instead of in S3: measured, no faster)
            if (AW && ta == 0) {
                // the windows before the first tile's
                const unsigned before4 = (static_cast<unsigned>(offset) << 2) - 256u;
                w_el = load4w(before4, 2, RingId<OALSFX_RV_EARLY_LINE>{}); w_lt = load4w(before4, 3, RingId<OALSFX_RV_MAIN>{}); w_ll = load4w(before4, 5, RingId<OALSFX_RV_LATE_LINE>{});
            }
            issue_taps_b(static_cast<unsigned>(offset + pos_a) << 2, tapbase(ta)); // on their way while the chain phases run
            __builtin_amdgcn_sched_barrier(0);
        }
        stamp();
        lds_barrier();
        stamp();
        // ---------------- S2: C1(ta), feedback half of the first shelf, beside C3(tb), first T60 section: 16 chains each, two wavefronts ----------------
        if (duty == ((NW == 2) ? 0 : 0 + ((NW > 4) ? (it & 1) * 4 : 0)) && chain_on && has_a) {
            __builtin_amdgcn_s_setprio(3); // the chain is one long dependency: let it issue ahead of the siblings' it phases on this SIMD
            float y1 = cdat[coop::LPY0], y2 = cdat[coop::LPY1];
            const float h1 = y1, h2 = y2;
            // (RG: a whole tile takes the loop with the constant trip count, as every tile of the other builds does: the variable one costs
            // the whole tiles of a ragged call a few per cent, and only the call's last tile needs it)
            if (!RG || La == 64) biquad_chain(crowI1, crowI2, 64, cdat[coop::LP_A1], cdat[coop::LP_A2], y1, y2);
            else biquad_chain(crowI1, crowI2, La, cdat[coop::LP_A1], cdat[coop::LP_A2], y1, y2);
            // history prefix for the second shelf's feed-forward half (behind the recurrence: its first requests do not wait for these stores)
            crowI2[3] = h1; crowI2[2] = h2;
            cdat[coop::LPY0] = y1; cdat[coop::LPY1] = y2;
            __builtin_amdgcn_s_setprio(0);
        }
        if (SF && any_sf && duty == 1 && has_c && lane < NW * kSfRows) {
            // SF: the first send shelves' recurrences of tile tc, lane = (instance, send, channel), on a wavefront that idles in S2
            const int cwx = lane / kSfRows, rr = lane % kSfRows, sd = rr / CH, c = rr % CH;
            float* m = sh.sf_misc[SF ? cwx : 0];
            const float* t = m + sfm::TAB + 12 * sd;
            const int mode = sh.sf_all[cwx] ? reinterpret_cast<const int*>(t)[10] : 0;
            if (mode & 4) {
                float* h = m + sfm::HIST + 6 * rr;
                float* wrow = sh.sf_rows[SF ? cwx : 0][SF ? rr : 0];
                const float* e = m + sfm::EDGE + 4 * c; // the tile's input samples 0, 1, 62, 63 of this channel
                float y1 = h[2], y2 = h[3];
                if (mode & 1) {
                    const float x1 = h[0], x2 = h[1];
                    wrow[4] = (t[0] * e[0]) + (t[1] * x1) + (t[2] * x2);
                    wrow[5] = (t[0] * e[1]) + (t[1] * e[0]) + (t[2] * x1);
                }
                wrow[2] = y2; wrow[3] = y1; // the second shelf's feed-forward sums read its input history here
                if (mode & 1) biquad_chain(wrow, wrow, 64, t[6], t[7], y1, y2);
                else { y1 = e[3]; y2 = e[2]; } // a shelf that is off passes its input through and lets its history follow (process_pass_through)
                h[0] = e[3]; h[1] = e[2]; h[2] = y1; h[3] = y2;
            }
        }
        if (duty == ((NW == 2) ? 0 : 2 + ((NW > 4) ? (it & 1) * 4 : 0)) && chain_on && has_b) {
            __builtin_amdgcn_s_setprio(3); // the chain is one long dependency: let it issue ahead of the siblings' it phases on this SIMD
            float prev = cdat[coop::T60O1];
            const float before = prev;
#ifndef OALSFX_ABLATE_T60_CHAINS // timing experiment (scripts/README: upper bound of what a parallel prefix of these sections could save; results wrong)
            if (!RG || Lb == 64) first_order_chain(crowL1, crowL2, 0, 64, cdat[coop::T_L2], 1.0F, false, prev);
            else first_order_chain(crowL1, crowL2, 0, Lb, cdat[coop::T_L2], 1.0F, false, prev);
#endif
            crowL2[3] = before; // the second section's feed-forward half needs o1[-1]
            cdat[coop::T60O1] = prev;
            __builtin_amdgcn_s_setprio(0);
        }
        stamp();
        lds_barrier();
        stamp();
#if OALSFX_ABLATE_VALU > 0
        // (ablation, profiles/r04c_instruction_diet/valu_ablation.txt: this many more vector instructions per wavefront and tile, doing
        // nothing -- does the step get longer by what they take to issue?)
        {
            // (eight independent registers in turn: issue slots, not a chain of latencies)
            float j0 = static_cast<float>(lane), j1 = j0 + 1.0F, j2 = j0 + 2.0F, j3 = j0 + 3.0F, j4 = j0 + 4.0F, j5 = j0 + 5.0F, j6 = j0 + 6.0F, j7 = j0 + 7.0F;
#pragma unroll
            for (int k = 0; k < OALSFX_ABLATE_VALU / 8; ++k)
                asm volatile("v_mul_f32 %0, %0, %0\n\tv_mul_f32 %1, %1, %1\n\tv_mul_f32 %2, %2, %2\n\tv_mul_f32 %3, %3, %3\n\t"
                             "v_mul_f32 %4, %4, %4\n\tv_mul_f32 %5, %5, %5\n\tv_mul_f32 %6, %6, %6\n\tv_mul_f32 %7, %7, %7"
                             : "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6), "+v"(j7));
            if (j0 + j1 + j2 + j3 + j4 + j5 + j6 + j7 == 12345.678F && ctx.timeline) ctx.timeline[1] = 1;
        }
#endif
        // ---------------- S3: P2(ta), feed-forward half of the second shelf; P4(tb), second T60 feed-forward ----------------
        if (go && has_a) {
            issue_taps_c(static_cast<unsigned>(offset + pos_a) << 2, tapbase(ta));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (SF && sf && has_c) {
            // SF: the second send shelves' feed-forward halves of tile tc (lane = frame): from the first shelves' rows to their own
#pragma unroll
            for (int sd = 0; sd < 2; ++sd) {
                const float* t = sfmisc + sfm::TAB + 12 * sd;
                const int mode = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(t)[10]);
                if (!(mode & 4)) continue;
#pragma unroll
                for (int c = 0; c < (MC ? 0 : CH); ++c) {
                    const float* y = sfrow(0, sd * CH + c) + 4 + lane;
                    float v = y[0];
                    if ((mode & 6) == 6) v = (t[3] * y[0]) + (t[4] * y[-1]) + (t[5] * y[-2]);
                    sfrow(1, sd * CH + c)[4 + lane] = v;
                }
            }
        }
        if (any_eax) {
            if (go && eax && has_a) {
                const v4f bq = *reinterpret_cast<const v4f*>(utf + ut::HPB);
                const float* xa = rowI(2, 0) + 4 + lane; const float* xb = rowI(2, 1) + 4 + lane;
                const float* xc = rowI(2, 2) + 4 + lane; const float* xd = rowI(2, 3) + 4 + lane;
                const v2f u01 = ((bq.x * v2f{xa[0], xb[0]}) + (bq.y * v2f{xa[-1], xb[-1]})) + (bq.z * v2f{xa[-2], xb[-2]});
                const v2f u23 = ((bq.x * v2f{xc[0], xd[0]}) + (bq.y * v2f{xc[-1], xd[-1]})) + (bq.z * v2f{xc[-2], xd[-2]});
                rowI(1, 0)[4 + lane] = u01.x; rowI(1, 1)[4 + lane] = u01.y; rowI(1, 2)[4 + lane] = u23.x; rowI(1, 3)[4 + lane] = u23.y;
            }
        }
        if (go && has_b) {
            const v4f c0 = *reinterpret_cast<const v4f*>(utf + ut::TH0);
            const v4f c1 = *reinterpret_cast<const v4f*>(utf + ut::TH1);
            const float* xa = rowL(2, 0) + 4 + lane; const float* xb = rowL(2, 1) + 4 + lane;
            const float* xc = rowL(2, 2) + 4 + lane; const float* xd = rowL(2, 3) + 4 + lane;
            const v2f w01 = (v2f{c0.x, c0.y} * v2f{xa[0], xb[0]}) + (v2f{c1.x, c1.y} * v2f{xa[-1], xb[-1]});
            const v2f w23 = (v2f{c0.z, c0.w} * v2f{xc[0], xd[0]}) + (v2f{c1.z, c1.w} * v2f{xc[-1], xd[-1]});
            rowL(1, 0)[4 + lane] = w01.x; rowL(1, 1)[4 + lane] = w01.y; rowL(1, 2)[4 + lane] = w23.x; rowL(1, 3)[4 + lane] = w23.y;
        }
        stamp();
        lds_barrier();
        stamp();
        // ---------------- S4: C2(ta) beside C4(tb), second T60 section and mid gain ----------------
        if (any_eax) {
            if (duty == ((NW == 2) ? 1 : 1 + ((NW > 4) ? (it & 1) * 4 : 0)) && chain2_on && has_a) {
                __builtin_amdgcn_s_setprio(3); // the chain is one long dependency: let it issue ahead of the siblings' it phases on this SIMD
                float y1 = cdat[coop::HPY0], y2 = cdat[coop::HPY1];
                if (!RG || La == 64) biquad_chain(crowI1, crowI0, 64, cdat[coop::HP_A1], cdat[coop::HP_A2], y1, y2);
                else biquad_chain(crowI1, crowI0, La, cdat[coop::HP_A1], cdat[coop::HP_A2], y1, y2);
                cdat[coop::HPY0] = y1; cdat[coop::HPY1] = y2;
                __builtin_amdgcn_s_setprio(0);
            }
        }
        if (SF && any_sf && duty == 0 && has_c && lane < NW * kSfRows) {
            // SF: the second send shelves' recurrences of tile tc, on a wavefront that idles in S4
            const int cwx = lane / kSfRows, rr = lane % kSfRows, sd = rr / CH;
            float* m = sh.sf_misc[SF ? cwx : 0];
            const float* t = m + sfm::TAB + 12 * sd;
            const int mode = sh.sf_all[cwx] ? reinterpret_cast<const int*>(t)[10] : 0;
            if (mode & 4) {
                float* h = m + sfm::HIST + 6 * rr;
                float z1 = h[4], z2 = h[5];
                float* zrow = sh.sf_rows[SF ? cwx : 0][SF ? kSfRows + rr : 0];
                if (mode & 2) biquad_chain(zrow, zrow, 64, t[8], t[9], z1, z2);
                else { z1 = h[2]; z2 = h[3]; }
                h[4] = z1; h[5] = z2;
            }
        }
        if (duty == ((NW == 2) ? 1 : 3 + ((NW > 4) ? (it & 1) * 4 : 0)) && chain_on && has_b) {
            __builtin_amdgcn_s_setprio(3); // the chain is one long dependency: let it issue ahead of the siblings' it phases on this SIMD
            float prev = cdat[coop::T60O2];
#ifndef OALSFX_ABLATE_T60_CHAINS
            if (!RG || Lb == 64) first_order_chain(crowL1, crowL1, 0, 64, cdat[coop::T_H2], cdat[coop::T_MID], true, prev);
            else first_order_chain(crowL1, crowL1, 0, Lb, cdat[coop::T_H2], cdat[coop::T_MID], true, prev);
#endif
            cdat[coop::T60O2] = prev;
            __builtin_amdgcn_s_setprio(0);
        }
        stamp();
        lds_barrier();
        stamp();
        // ---------------- S5: P5(tb): late all-pass, ring writes, outputs ----------------
        if (EH && it == tiles) hand_back();
        if (go && has_b) {
            const v2f i01 = {rowL(1, 0)[4 + lane], rowL(1, 1)[4 + lane]};
            const v2f i23 = {rowL(1, 2)[4 + lane], rowL(1, 3)[4 + lane]};
            // XF, a cross-fading tile: the late all-pass taps being faded in are requested here (nothing of this tile has touched that ring yet)
            v4f q_la = {0, 0, 0, 0};
            if (XF && xf_faded) q_la = load4(t4, 4, RingId<OALSFX_RV_LATE_AP>{}, kTapN);
            if (ST && (short_mask & 16u)) {
                // late all-pass offsets shorter than the tile, as for the early all-pass
                const v4u d = *reinterpret_cast<const v4u*>(utu + (XF ? tb_base : static_cast<int>(ut::TAP4)) + 16);
                const int o0x = static_cast<int>(d.x >> 2), o1x = static_cast<int>(d.y >> 2), o2x = static_cast<int>(d.z >> 2), o3x = static_cast<int>(d.w >> 2);
                int shortest_off = min(min(o0x, o1x), min(o2x, o3x));
                const v4u dn = *reinterpret_cast<const v4u*>(utu + (XF ? kTapN : static_cast<int>(ut::TAP4)) + 16);
                const int n0x = static_cast<int>(dn.x >> 2), n1x = static_cast<int>(dn.y >> 2), n2x = static_cast<int>(dn.z >> 2), n3x = static_cast<int>(dn.w >> 2);
                if (XF && xf_faded) shortest_off = min(shortest_off, min(min(n0x, n1x), min(n2x, n3x)));
                const int ahead = 63 / shortest_off;
                for (int k = 0; k < ahead; ++k) {
                    v4f a_now = p_la;
                    if (XF && xf_faded) mix4(a_now, q_la);
                    const v2f pl01 = v2f{a_now.x, a_now.y} - (ac * i01);
                    const v2f pl23 = v2f{a_now.z, a_now.w} - (ac * i23);
                    v2f pq01 = i01 + (ac * pl01), pq23 = i23 + (ac * pl23);
                    scatter2(pq01, pq23, sx, sy);
                    strow(0)[4 + lane] = pq01.x; strow(1)[4 + lane] = pq01.y; strow(2)[4 + lane] = pq23.x; strow(3)[4 + lane] = pq23.y;
                    wave_sync();
                    if (lane >= o0x) p_la.x = strow(0)[4 + lane - o0x];
                    if (lane >= o1x) p_la.y = strow(1)[4 + lane - o1x];
                    if (lane >= o2x) p_la.z = strow(2)[4 + lane - o2x];
                    if (lane >= o3x) p_la.w = strow(3)[4 + lane - o3x];
                    if (XF && xf_faded) {
                        if (lane >= n0x) q_la.x = strow(0)[4 + lane - n0x];
                        if (lane >= n1x) q_la.y = strow(1)[4 + lane - n1x];
                        if (lane >= n2x) q_la.z = strow(2)[4 + lane - n2x];
                        if (lane >= n3x) q_la.w = strow(3)[4 + lane - n3x];
                    }
                    wave_sync();
                }
            }
            if (XF && xf_faded) mix4(p_la, q_la);
            const v2f l01 = v2f{p_la.x, p_la.y} - (ac * i01);
            const v2f l23 = v2f{p_la.z, p_la.w} - (ac * i23);
            v2f q01 = i01 + (ac * l01), q23 = i23 + (ac * l23);
            scatter2(q01, q23, sx, sy);
            if constexpr (CR == 2) {
                const int f4 = cr_lane_before(utu[ut::CR_R]);
                carry_store_lds(tile_b4, RingId<OALSFX_RV_LATE_AP>{}, 3, v4f{cr_rot(f4, q01.x), cr_rot(f4, q01.y), cr_rot(f4, q23.x), cr_rot(f4, q23.y)}, head_b, tail_b, Lb);
            } else if (act) store4(t4, RingId<OALSFX_RV_LATE_AP>{}, q01.x, q01.y, q23.x, q23.y);
            v2f r01 = {l23.y, l23.x}, r23 = {l01.y, l01.x};
            scatter2(r01, r23, sx, sy);
            if constexpr (CR == 2) {
                const int f4 = cr_lane_before(utu[ut::CR_R]);
                carry_store_lds(tile_b4, RingId<OALSFX_RV_LATE_LINE>{}, 4, v4f{cr_rot(f4, r01.x), cr_rot(f4, r01.y), cr_rot(f4, r23.x), cr_rot(f4, r23.y)}, head_b, tail_b, Lb);
            } else if (act) store4(t4, RingId<OALSFX_RV_LATE_LINE>{}, r01.x, r01.y, r23.x, r23.y);
            const float data[8] = {e01.x, e01.y, e23.x, e23.y, l01.x, l01.y, l23.x, l23.y};
            if (MC) {
                // early lines 0..3 then late lines 0..3, each into every audible channel (reference src/oalsfxpp.cpp:6142-6166)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (aud_out & (1ULL << (k * 8 + c))) outv[MC ? c : 0] += data[k] * utf[kMcBase + k * 8 + c];
                }
                if (!act) {
                    // a lane past the end of a ragged call's last tile
                } else if (last && ctx.turn_set != 0u) {
                    // a chained launch writes the caller's frames through, as the stereo builds do (see below): two channels a store where a
                    // frame is an even number of channels (quad, 5.1, 7.1); 6.1's seven below (one store apiece measured slower than stream order,
                    // 86.5 against 74.6 us per step)
                    if ((nch & 1) == 0) {
#pragma unroll
                        for (int c = 0; c < 8; c += 2)
                            if (c < nch) {
                                const unsigned long long both = static_cast<unsigned long long>(__float_as_uint(outv[MC ? c : 0])) |
                                                                (static_cast<unsigned long long>(__float_as_uint(outv[MC ? c + 1 : 0])) << 32);
                                __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst + static_cast<size_t>(pos_b) * nch + c), both, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                    } else {
                        // an odd number of channels (6.1: seven): a pair must start at an even float, which depends on the frame's parity -- an
                        // even frame stores (0,1) (2,3) (4,5) and its last channel alone, an odd one its first channel alone and (1,2) (3,4) (5,6)
                        const bool odd = (pos_b & 1) != 0;
                        float* frame = dst + static_cast<size_t>(pos_b) * nch;
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const float lo = odd ? outv[MC ? 2 * k + 1 : 0] : outv[MC ? 2 * k : 0];
                            const float hi = odd ? outv[MC ? 2 * k + 2 : 0] : outv[MC ? 2 * k + 1 : 0];
                            const int at = 2 * k + (odd ? 1 : 0);
                            if (at + 1 < nch) {
                                const unsigned long long both = static_cast<unsigned long long>(__float_as_uint(lo)) | (static_cast<unsigned long long>(__float_as_uint(hi)) << 32);
                                __hip_atomic_store(reinterpret_cast<unsigned long long*>(frame + at), both, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                        // (nch is 7 here -- the only odd count above two -- but written for any: the channel no pair took)
                        const int alone = odd ? 0 : nch - 1;
                        float v = outv[0];
#pragma unroll
                        for (int c = 1; c < 8; ++c)
                            if (MC && c == alone) v = outv[MC ? c : 0];
                        __hip_atomic_store(reinterpret_cast<unsigned*>(frame + alone), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                } else if (last) {
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (c < nch) dst[static_cast<size_t>(pos_b) * nch + c] = outv[MC ? c : 0];
                } else {
#pragma unroll
                    for (int c = 0; c < 8; ++c)
                        if (c < nch) mixbuf[c * OALSFX_MAX_CHUNK + pos_b] = outv[MC ? c : 0];
                }
            }
            if (XF && xf_active) {
                const int tpos = tb << 6;
                if (tpos == blk_end) {
                    // a block starts with this tile
                    blk_start = tpos;
                    int todo = min(frames - tpos, OALSFX_RV_MAX_UPDATE);
                    if (fc < OALSFX_RV_FADE_SAMPLES) todo = min(todo, OALSFX_RV_FADE_SAMPLES - fc);
                    blk_end = tpos + todo;
                    blk_counter = frames - tpos;
                    g_step = (g_tgt - g_cur) * (1.0F / static_cast<float>(blk_counter));
                    g_ramp = q_valid && (fabsf(g_step) > FLT_EPSILON);
                    ramp_mask = __ballot(g_ramp);
                    g_run = g_cur;
                }
                if (ramp_mask != 0ULL) {
                    if (g_ramp) {
                        float* gs = grow(lane);
                        for (int i = 0; i < 64; ++i) {
                            gs[i] = g_run;
                            g_run += g_step;
                        }
                    }
                    wave_sync();
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
#pragma unroll
                    for (int c = 0; c < (MC ? 0 : CH); ++c) {
                        const int q = k * CH + c;
                        float& o = c == 0 ? o0 : o1;
                        if ((ramp_mask >> q) & 1ULL) {
                            o += data[k] * grow(q)[lane];
                        } else {
                            const float gq = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(g_cur), q));
                            if (audible(gq)) o += data[k] * gq;
                        }
                    }
                }
                if (ramp_mask != 0ULL) wave_sync(); // the rows go back to the late half
                if (tpos + 64 == blk_end) {
                    // the block ends with this tile (a ramp that reaches the end of the call lands on its target exactly)
                    if (g_ramp) g_cur = (blk_end == frames) ? g_tgt : g_run;
                    if (fc < OALSFX_RV_FADE_SAMPLES) fc = min(fc + (blk_end - blk_start), OALSFX_RV_FADE_SAMPLES);
                }
            } else {
#pragma unroll
                for (int k = 0; k < (MC ? 0 : 8); k += 2) {
                    const v4f g = *reinterpret_cast<const v4f*>(utf + ut::GOUT + 2 * k);
                    if (aud_out & (1u << (2 * k + 0))) o0 += data[k] * g.x;
                    if (CH == 2 && (aud_out & (1u << (2 * k + 1)))) o1 += data[k] * g.y;
                    if (aud_out & (1u << (2 * k + 2))) o0 += data[k + 1] * g.z;
                    if (CH == 2 && (aud_out & (1u << (2 * k + 3)))) o1 += data[k + 1] * g.w;
                }
            }
            if (MC || !act) {
                // stored above / a lane past the end of a ragged call's last tile
            } else if (last) {
                if (ctx.turn_set != 0u && !(OALSFX_CHAIN_EXP & 2)) {
                    // chained launches: the caller's buffer is ordinary memory, and two launches may write the same frames from two XCDs:
                    // written through (agent scope), so that no older line waits in another L2 to be written back over this one
                    if (CH == 2) {
                        const unsigned long long both = static_cast<unsigned long long>(__float_as_uint(o0)) | (static_cast<unsigned long long>(__float_as_uint(o1)) << 32);
                        __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst + static_cast<size_t>(pos_b) * 2), both, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else __hip_atomic_store(reinterpret_cast<unsigned*>(dst + pos_b), __float_as_uint(o0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else if (CH == 2) {
                    if (OALSFX_NT & 16) __builtin_nontemporal_store(v2f{o0, o1}, reinterpret_cast<v2f*>(dst + static_cast<size_t>(pos_b) * 2));
                    else *reinterpret_cast<float2*>(dst + static_cast<size_t>(pos_b) * 2) = make_float2(o0, o1);
                } else dst[pos_b] = o0;
            } else {
                mixbuf[pos_b] = o0;
                if (CH == 2) mixbuf[OALSFX_MAX_CHUNK + pos_b] = o1;
            }
            wave_sync(); // ring stores of this tile precede the loads of the tile after next (program order)
        }
        o0 = oa0; o1 = oa1;
#pragma unroll
        for (int c = 0; c < (MC ? 8 : 1); ++c) outv[c] = outva[c];
        stamp();
    }

    // ---- hand the state back ----
    if (!EH) hand_back();
    if constexpr (FP) {
    } else
    // ---- an instance that is not in its steady state after all (the host only guesses): the general path, out of line ----
    if constexpr (NF) {
        // NF: a build without the general path inside; the host lists only instances whose test it can predict from what it knows
        // (DESIGN 4), and one that fails it anyway is counted like in an FP build
        if (valid && lane == 0 && !go && ctx.fault) __hip_atomic_fetch_add(ctx.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if (MC || ctx.progress != nullptr) {
        // the general kernel follows on the same list: tell it how far this instance got
        if (valid && lane == 0) ctx.progress[sidx] = go ? frames : 0;
        if (valid && lane == 0 && !go && w < ctx.no_follow_up && ctx.fault) __hip_atomic_fetch_add(ctx.fault, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else if constexpr (!MC) {
        if (valid && !go) {
            KernelCtx copy = ctx; // a copy made here only: taking the address of the kernel argument itself would park it in scratch for every wave
            reverb_general_call<CH>(&copy, slot, inst, flags & 0xFF, lds, lane);
        }
    }
    if (ctx.turn != nullptr && ctx.turn_set != 0u) {
        // this launch is through with the instance (its stores acknowledged): the next one may take it
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (valid && lane == 0) { // (where: see the wait)
            __hip_atomic_store(ctx.turn_cu2 + sidx, cu_before, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(ctx.turn_cu + sidx, this_cu(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_waitcnt(0); // vmcnt(0) expcnt(0) lgkmcnt(0)
#if OALSFX_CHAIN_EXP & 4
        // (experiment, ADVICE round 3: the word as an agent-scope release store -- buffer_wbl2 sc1 in front of it, a write-back of this
        // XCD's L2 per wavefront; what launches hand on is uncached and has nothing dirty there, the caller's output frames are written
        // through: measured, profiles/r04c_instruction_diet/release_store_ab.txt)
        if (valid && lane == 0) __hip_atomic_store(ctx.turn + sidx, ctx.turn_set, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
#else
        if (valid && lane == 0) __hip_atomic_store(ctx.turn + sidx, ctx.turn_set, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    }
    stamp(); // state handed back
}

template <int CH, int NW, bool TL = false, bool HY = false, bool MD = false, bool ST = false, bool RG = false, bool FP = false, bool XF = false, bool NF = false, bool SF = false, int CR = 0>
__global__ __launch_bounds__(64 * NW, 4) void k_reverb_steady_coop(KernelCtx ctx, int slot, const int* __restrict__ list, int count, int flags)
{
    __shared__ SteadyShared<CH, NW, FP, MD, ST, SF, CR> sh;
    reverb_steady_group<CH, NW, TL, HY, MD, ST, RG, FP, XF, NF, SF, CR>(ctx, slot, list, count, flags, static_cast<int>(blockIdx.x), sh);
}

// The same builds taking exactly 128 registers per lane whatever they need (OALSFX_EQUAL_PLACES, common.hpp): for a chained step of two
// different kernels (batch.cpp), whose workgroups must fit each other's places.  Not for a run of this kernel alone: with every
// wavefront at 128 registers a full chip has none left, and the one-wavefront gate in front of the next launch (k_chain_gate) finds no
// place until a workgroup leaves -- the headline measured 41.3 us per step that way against 40.6 with the build's own 120, the
// driver's 20-step command 45.3 against 43.7 (profiles/r04n_places_and_the_gate/).
template <int CH, int NW, bool TL = false, bool HY = false, bool MD = false, bool ST = false, bool RG = false, bool FP = false, bool XF = false, bool NF = false, bool SF = false, int CR = 0>
__global__ __launch_bounds__(64 * NW, 4) void k_reverb_steady_coop_ep(KernelCtx ctx, int slot, const int* __restrict__ list, int count, int flags)
{
    __shared__ SteadyShared<CH, NW, FP, MD, ST, SF, CR> sh;
    reverb_steady_group<CH, NW, TL, HY, MD, ST, RG, FP, XF, NF, SF, CR>(ctx, slot, list, count, flags, static_cast<int>(blockIdx.x), sh);
    if constexpr (!(RG && CR == 2 && CH == 2)) OALSFX_EQUAL_PLACES(); // (that build takes 128 registers as it is, and spilled with the statement)
}

// One grid for a slot's steady reverbs of several kinds (mono / stereo, whole tiles).  The host orders the slot's list by kind and the
// workgroups take the build of theirs: proven instances whose taps are all two tiles away (plain FP), proven ones with a tap of one to
// two tiles (HY FP), proven ones with shorter taps or a modulated late line (the most general FP build), and the instances that are
// believed steady or in a transition the XF build follows.  One property change, or one preset with a short tap, among thousands of
// instances then costs its own workgroup the slower build and nobody else.  (NF: the last kind without the general path inside.)
struct SteadyKinds {
    int count[4]; // plain, close taps, short taps or modulated, believed / in transition: list entries, in this order; every kind starts a new workgroup
    __host__ __device__ int groups(int k) const { return (count[k] + 3) >> 2; }
};

// (plain FP workgroups of a grid of kinds: the late feed's stores line-aligned unless the send filters are inside, whose LDS leaves no
// room to choose; CR == 2: every store site, the build for write positions off the line grid)
constexpr int kCrBase = OALSFX_CR_FEED ? 1 : 0;
template <int CH, bool NF, bool SF, int CR = (SF ? 0 : kCrBase)>
__global__ __launch_bounds__(256, 4) void k_reverb_steady_kinds(KernelCtx ctx, int slot, const int* __restrict__ list, SteadyKinds kinds, int flags)
{
    static_assert(!SF || CR == 0, "send filters inside and line-aligned stores: no LDS for both");
    union Shared {
        SteadyShared<CH, 4, true, false, false, SF, CR> lean; // plain and HY (SF: with the send filters inside)
        SteadyShared<CH, 4, true, true, true> general;   // ST (includes MD)
        SteadyShared<CH, 4, false, true, true> believed; // XF
    };
    __shared__ Shared sh;
    // The workgroups of the last kind -- believed or in a transition: the slowest build, 65 us per 256 frames where the proven ones take
    // 40 -- are the grid's first: a launch dispatches its workgroups in order, and the ones that take longest should not be the ones that
    // start last (with consecutive launches overlapping, a workgroup starts when a slot of the launch before is given up).
    const int slow_groups = kinds.groups(3);
    if (static_cast<int>(blockIdx.x) < slow_groups) {
        reverb_steady_group<CH, 4, false, true, true, true, false, false, true, NF>(ctx, slot, list + kinds.count[0] + kinds.count[1] + kinds.count[2], kinds.count[3], flags,
                                                                                     static_cast<int>(blockIdx.x), sh.believed);
        return;
    }
    // the others in CU-major order: the workgroups that share a CU run the same build (as far as the kinds' sizes allow)
    const int rest = static_cast<int>(blockIdx.x) - slow_groups;
    int group = (flags & kNoCuMajor) ? rest : cu_major_position(rest, static_cast<int>(gridDim.x) - slow_groups);
    if (group < kinds.groups(0)) {
        reverb_steady_group<CH, 4, false, false, false, false, false, true, false, false, SF, CR>(ctx, slot, list, kinds.count[0], flags, group, sh.lean);
        return;
    }
    group -= kinds.groups(0);
    list += kinds.count[0];
    if (group < kinds.groups(1)) {
        reverb_steady_group<CH, 4, false, true, false, false, false, true, false, false, SF>(ctx, slot, list, kinds.count[1], flags, group, sh.lean);
        return;
    }
    group -= kinds.groups(1);
    list += kinds.count[1];
    reverb_steady_group<CH, 4, false, true, true, true, false, true>(ctx, slot, list, kinds.count[2], flags, group, sh.general);
}

// General path for one instance on one wavefront: any cross-fade state, modulation, gain ramps, taps closer than a tile,
// partial tiles, any channel count.
template <int CH>
__device__ __forceinline__ void reverb_general_instance(const KernelCtx& ctx, int slot, int inst, int flags, float* lds, int lane)
{
    constexpr int NQ = Lds<CH>::kChains;
    auto row = [&](int group, int c) -> float* { return lds + (group * 4 + c) * kRow; };
    float* rng = lds + kGroups * 4 * kRow;
    float* gseq = rng + kRngFloats;
    float* utf = gseq + NQ * 64;                             // uniform table, float view
    unsigned* utu = reinterpret_cast<unsigned*>(utf);        // ... unsigned view

    // measurement only (OALSFX_DEBUG_TIMELINE): every 64th instance stamps the shader clock at its phase boundaries, into the
    // second half of the timeline buffer
    int gts_i = 0;
    auto gstamp = [&]() {
        if (ctx.timeline && (inst & 63) == 0 && (inst >> 6) < 64 && lane == 0 && gts_i < 96)
            ctx.timeline[64 * 4 * 96 + (inst >> 6) * 96 + gts_i++] = clock64();
    };
    gstamp();
    const int channels = (CH == 8) ? ctx.channels : CH;
    const int frames = ctx.frames;
    const size_t sidx = static_cast<size_t>(inst) * ctx.slots + slot;
    // frames of this chunk the steady-state kernel has done already (launched right before on the same list), else 0
    const int resume = ctx.progress ? __builtin_amdgcn_readfirstlane(ctx.progress[sidx]) : 0;
    if (resume >= frames) return;
    // parameters are read-only for the whole launch: address space 4 (constant) makes every access a scalar load
    typedef const __attribute__((address_space(4))) oalsfx_slot_params ConstSlotParams;
    typedef const __attribute__((address_space(4))) oalsfx_source_params ConstSourceParams;
    ConstSlotParams& SP = *(ConstSlotParams*)(uintptr_t)(ctx.params + sidx);
    const auto& P = SP.u.reverb;
    oalsfx_slot_state& SS = ctx.state[sidx];
    oalsfx_reverb_state& S = SS.u.reverb;
    ConstSourceParams& SRC = *(ConstSourceParams*)(uintptr_t)(ctx.source + inst);
    float* slab = ctx.rings[sidx];
    const bool first = (flags & kFirst) != 0;
    const bool last = (flags & kLast) != 0;
    const bool eax = P.is_eax != 0;

    GlobalBytes* slab_b = (GlobalBytes*)(uintptr_t)slab;
    Ring ring[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        ring[r].bmask = static_cast<unsigned>(P.ring_len[r] - 1) << 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) ring[r].lo[j] = static_cast<unsigned>(P.ring_off[r] + j * P.ring_len[r]) << 2;
    }
    const Ring& r_main = ring[OALSFX_RV_MAIN];
    const Ring& r_eap = ring[OALSFX_RV_EARLY_AP];
    const Ring& r_eline = ring[OALSFX_RV_EARLY_LINE];
    const Ring& r_lap = ring[OALSFX_RV_LATE_AP];
    const Ring& r_lline = ring[OALSFX_RV_LATE_LINE];

    // ---- instance state into registers (all wave-uniform) ----
    int fade_count = S.fade_count, offset = S.offset, mod_index = S.mod_index, mod_range = S.mod_range;
    float mod_filter = S.mod_filter;

    // pending parameter update: what do_update does to state (reference src/oalsfxpp.cpp:7028-7031, 6062-6075)
    if (SS.seen_seq != SP.update_seq) {
        mod_index = static_cast<int>(static_cast<long long>(mod_index) * P.mod_range / mod_range);
        mod_range = P.mod_range;
        bool differ = false;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            differ |= (P.early_tap[j] != S.cur_early_tap[j]) | (P.early_ap_off[j] != S.cur_early_ap_off[j]) |
                      (P.early_line_off[j] != S.cur_early_line_off[j]) | (P.late_tap[j] != S.cur_late_tap[j]) |
                      (P.late_ap_off[j] != S.cur_late_ap_off[j]) | (P.late_line_off[j] != S.cur_late_line_off[j]);
        if (differ) fade_count = 0;
    }

    // ---- per-wave table of instance constants in LDS (see namespace ut) ----
    // the six current-tap arrays are contiguous in oalsfx_reverb_state, in the table's group order
    if (lane < 24) utu[ut::TAP4 + lane] = 4u * static_cast<unsigned>((&S.cur_early_tap[0])[lane]);
    if (lane < 20) {
        const oalsfx_reverb_params& PG = ctx.params[sidx].u.reverb;
        const int r = lane >> 2, j = lane & 3;
        utu[ut::LO + lane] = static_cast<unsigned>(PG.ring_off[r] + j * PG.ring_len[r]) << 2;
        if (lane < 5) utu[ut::BMASK + lane] = static_cast<unsigned>(PG.ring_len[lane] - 1) << 2;
        if (lane < 4) {
            utf[ut::ECOEF + lane] = PG.early_tap_coeff[lane];
            utf[ut::ELCOEF + lane] = PG.early_line_coeff[lane];
            utf[ut::TL0 + lane] = PG.t60_lf[lane][0];
            utf[ut::TL1 + lane] = PG.t60_lf[lane][1];
            utf[ut::TH0 + lane] = PG.t60_hf[lane][0];
            utf[ut::TH1 + lane] = PG.t60_hf[lane][1];
        }
        if (CH <= 2) {
            const oalsfx_source_params& SG = ctx.source[inst];
            if (lane < 4) utf[ut::GDIR + lane] = SG.direct.gains[lane >> 1][lane & 1];
            if (lane < 8) utf[ut::GAUX + lane] = SG.aux[slot].gains[lane >> 2][lane & 3];
        }
    }
    if (lane == 0) {
        utu[ut::FEED4] = 4u * static_cast<unsigned>(P.late_feed_tap);
        utf[ut::MISC + 0] = P.density_gain; utf[ut::MISC + 1] = P.ap_feed_coeff; utf[ut::MISC + 2] = P.mix_x; utf[ut::MISC + 3] = P.mix_y;
        utf[ut::LPB + 0] = P.lp.b0; utf[ut::LPB + 1] = P.lp.b1; utf[ut::LPB + 2] = P.lp.b2; utf[ut::LPB + 3] = 0.0F;
        utf[ut::HPB + 0] = P.hp.b0; utf[ut::HPB + 1] = P.hp.b1; utf[ut::HPB + 2] = P.hp.b2; utf[ut::HPB + 3] = 0.0F;
    }
    // audible send gains as bit masks (stereo / mono): bit c*2+o for the dry mix, bit c*4+k for the B-format send
    unsigned aud_dir = 0, aud_aux = 0;
    if (CH <= 2) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
#pragma unroll
            for (int o = 0; o < CH; ++o) aud_dir |= audible(SRC.direct.gains[c][o]) ? 1u << (c * 2 + o) : 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) aud_aux |= audible(SRC.aux[slot].gains[c][k]) ? 1u << (c * 4 + k) : 0u;
        }
    }
    wave_sync();

    // ---- chain lanes (0..3, one per line): filter histories in registers ----
    const int cl = lane & 3;
    float lpx0 = S.lp[cl].x[0], lpx1 = S.lp[cl].x[1], lpy0 = S.lp[cl].y[0], lpy1 = S.lp[cl].y[1];
    float hpy0 = S.hp[cl].y[0], hpy1 = S.hp[cl].y[1];
    float t60_x = S.t60[cl][0][0], t60_o1 = S.t60[cl][0][1], t60_o1b = S.t60[cl][1][0], t60_o2 = S.t60[cl][1][1];
    const float lp_a1 = P.lp.a1, lp_a2 = P.lp.a2, hp_a1 = P.hp.a1, hp_a2 = P.hp.a2;
    const float t_l2 = P.t60_lf[cl][2], t_h2 = P.t60_hf[cl][2], t_mid = P.t60_mid[cl];

    // ---- chain lanes: output gain ramps. chain q = (stage*4 + line)*CH + channel ----
    const int q_stage = lane / (4 * CH), q_line = (lane / CH) & 3, q_chan = lane % CH;
    const bool q_valid = (lane < NQ) && (q_chan < channels);
    float g_cur = 0.f, g_tgt = 0.f;
    if (q_valid) {
        g_cur = q_stage ? S.late_cur_gain[q_line][q_chan] : S.early_cur_gain[q_line][q_chan];
        g_tgt = q_stage ? P.late_pan[q_line][q_chan] : P.early_pan[q_line][q_chan];
    }

    const bool filtered = (flags & kFiltered) != 0 && instance_has_send_filter(ctx, inst);
    const float* src = ctx.raw_src + static_cast<size_t>(inst) * ctx.io_stride;
    const float* wsrc = src;
    if (filtered) {
        src = ctx.src + static_cast<size_t>(inst) * ctx.src_stride;
        wsrc = ctx.wet_src + static_cast<size_t>(inst) * ctx.src_stride;
    }
    float* dst = ctx.dst + static_cast<size_t>(inst) * ctx.io_stride;
    float* mixbuf = ctx.mixbuf ? ctx.mixbuf + static_cast<size_t>(inst) * channels * OALSFX_MAX_CHUNK : nullptr;

    const float b2a = 0.288675134595F; // |b2a| entries (reference src/oalsfxpp.cpp:6377-6383), signs applied below
    const float apc = P.ap_feed_coeff, mx = P.mix_x, my = P.mix_y;

    gstamp(); // prologue done
    for (int base = resume; base < frames;) {
        // the reference's blocks start every 256 frames of the call; carrying on behind the steady-state kernel may start in
        // the middle of one (at a multiple of 64), and the ramp counter still refers to the block's own start
        const int block_start = (base == resume) ? base - (base % OALSFX_RV_MAX_UPDATE) : base;
        int todo = min(frames - base, OALSFX_RV_MAX_UPDATE - (base - block_start));
        if (OALSFX_RV_FADE_SAMPLES - fade_count > 0) todo = min(todo, OALSFX_RV_FADE_SAMPLES - fade_count);
        const bool faded = fade_count < OALSFX_RV_FADE_SAMPLES; // fade < 1.0
        const int counter = frames - block_start;

        // output gain ramps of this chunk (MixHelpers::mix, reference src/oalsfxpp.cpp:2762-2786)
        const float delta = 1.0F / static_cast<float>(counter);
        const float g_step = (g_tgt - g_cur) * delta;
        const bool g_ramp = q_valid && (fabsf(g_step) > FLT_EPSILON);
        const unsigned long long ramp_mask = __ballot(g_ramp);
        float g_run = g_cur;

        // modulation depth smoother: a serial lerp chain over the chunk (reference src/oalsfxpp.cpp:7462)
        const bool mod_active = (P.mod_depth != 0.0F) || (mod_filter != 0.0F);
        if (mod_active) {
            if (lane == 0) {
                float r = mod_filter;
                for (int i = 0; i < todo; ++i) {
                    r = lerpf(r, P.mod_depth, P.mod_coeff);
                    rng[i] = r;
                }
            }
            wave_sync();
        }

        // current taps of this chunk (wave-uniform; they only change when a cross-fade completes)
        int cur_etap[4], cur_eap[4], cur_eline[4], cur_ltap[4], cur_lap[4], cur_lline[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            cur_etap[j] = __builtin_amdgcn_readfirstlane(utu[ut::TAP4 + 0 + j] >> 2);
            cur_eap[j] = __builtin_amdgcn_readfirstlane(utu[ut::TAP4 + 4 + j] >> 2);
            cur_eline[j] = __builtin_amdgcn_readfirstlane(utu[ut::TAP4 + 8 + j] >> 2);
            cur_ltap[j] = __builtin_amdgcn_readfirstlane(utu[ut::TAP4 + 12 + j] >> 2);
            cur_lap[j] = __builtin_amdgcn_readfirstlane(utu[ut::TAP4 + 16 + j] >> 2);
            cur_lline[j] = __builtin_amdgcn_readfirstlane(utu[ut::TAP4 + 20 + j] >> 2);
        }
        // early / late feedback limits: the shortest positive all-pass delay in use
        int eap_limit = 64, lap_limit = 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            eap_limit = min_positive(eap_limit, cur_eap[j]);
            lap_limit = min_positive(lap_limit, cur_lap[j]);
            if (faded) {
                eap_limit = min_positive(eap_limit, P.early_ap_off[j]);
                lap_limit = min_positive(lap_limit, P.late_ap_off[j]);
            }
        }
        // which tap groups lie entirely before a full tile and can be fetched up front (while cross-fading: the taps faded in as well)
        auto far = [&](const int* cur, const int* target, int least) { return min4(cur) >= least && (!faded || min4(target) >= least); };
        int n_etap[4], n_eap[4], n_eline[4], n_ltap[4], n_lap[4], n_lline[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            n_etap[j] = P.early_tap[j]; n_eap[j] = P.early_ap_off[j]; n_eline[j] = P.early_line_off[j];
            n_ltap[j] = P.late_tap[j]; n_lap[j] = P.late_ap_off[j]; n_lline[j] = P.late_line_off[j];
        }
        const bool pre_e = far(cur_etap, n_etap, 64);
        const bool pre_a = far(cur_eap, n_eap, 64);
        const bool pre_el = far(cur_eline, n_eline, 64);
        const bool pre_lt = far(cur_ltap, n_ltap, P.late_feed_tap + 64);
        const bool pre_la = far(cur_lap, n_lap, 64);
        const bool pre_ll = pre_la && !mod_active && far(cur_lline, n_lline, 64);
        for (int done = 0; done < todo; done += 64) {
            const int L = min(64, todo - done);
            const bool act = lane < L;
            const int t = offset + done + lane;                 // absolute sample index of this lane
            const unsigned t4 = static_cast<unsigned>(t) << 2;    // ... as a byte position in a float ring
            const int pos = base + done + lane;                 // index inside the caller's chunk
            const float fade = static_cast<float>(fade_count + done + lane) * (1.0F / OALSFX_RV_FADE_SAMPLES);

            wave_sync(); // ring stores of the previous tile precede the loads below (program order)

            gstamp(); // tile start
            // ---------------- loads that do not depend on this tile ----------------
            float in[CH];   // the frame as the direct send sees it
            float win[CH];  // ... as this slot's auxiliary send sees it (differs only after the send-filter pre-pass)
            float out[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) { in[c] = 0.0F; out[c] = 0.0F; }
            if (act) {
                if (CH == 2) {
                    const float2 v = *reinterpret_cast<const float2*>(src + static_cast<size_t>(pos) * 2);
                    in[0] = v.x; in[CH - 1] = v.y;
                } else {
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        if (c < channels) in[c] = src[static_cast<size_t>(pos) * channels + c];
                }
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) win[c] = in[c];
            if (filtered && act) {
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    if (c < channels) win[c] = wsrc[static_cast<size_t>(pos) * channels + c];
            }
            float p_e[4], p_a[4], p_el[4], p_lt[4], p_ll[4], p_la[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                p_e[j] = p_a[j] = p_el[j] = p_lt[j] = p_ll[j] = p_la[j] = 0.0F;
                if (act) {
                    if (pre_e) p_e[j] = ld(slab_b, r_main.at(j, t4 - 4u * cur_etap[j]));
                    if (pre_a) p_a[j] = ld(slab_b, r_eap.at(j, t4 - 4u * cur_eap[j]));
                    if (pre_el) p_el[j] = ld(slab_b, r_eline.at(j, t4 - 4u * cur_eline[j]));
                    if (pre_lt) p_lt[j] = ld(slab_b, r_main.at(j, t4 - 4u * cur_ltap[j]));
                    if (pre_ll) p_ll[j] = ld(slab_b, r_lline.at(j, t4 - 4u * cur_lline[j]));
                    if (pre_la) p_la[j] = ld(slab_b, r_lap.at(j, t4 - 4u * cur_lap[j]));
                }
            }
            if (faded) {
                // cross-fading: the taps being faded in travel with them, and the two are mixed here as delay_out_faded does
                // (reference src/oalsfxpp.cpp:7358-7399), one round trip for the tile instead of one in front of every stage
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float q_e = 0.0F, q_a = 0.0F, q_el = 0.0F, q_lt = 0.0F, q_ll = 0.0F, q_la = 0.0F;
                    if (act) {
                        if (pre_e) q_e = ld(slab_b, r_main.at(j, t4 - 4u * n_etap[j]));
                        if (pre_a) q_a = ld(slab_b, r_eap.at(j, t4 - 4u * n_eap[j]));
                        if (pre_el) q_el = ld(slab_b, r_eline.at(j, t4 - 4u * n_eline[j]));
                        if (pre_lt) q_lt = ld(slab_b, r_main.at(j, t4 - 4u * n_ltap[j]));
                        if (pre_ll) q_ll = ld(slab_b, r_lline.at(j, t4 - 4u * n_lline[j]));
                        if (pre_la) q_la = ld(slab_b, r_lap.at(j, t4 - 4u * n_lap[j]));
                    }
                    p_e[j] = lerpf(p_e[j], q_e, fade); p_a[j] = lerpf(p_a[j], q_a, fade); p_el[j] = lerpf(p_el[j], q_el, fade);
                    p_lt[j] = lerpf(p_lt[j], q_lt, fade); p_ll[j] = lerpf(p_ll[j], q_ll, fade); p_la[j] = lerpf(p_la[j], q_la, fade);
                }
            }
            if (!first && act) {
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    if (c < channels) out[c] = mixbuf[c * OALSFX_MAX_CHUNK + pos];
            }

            gstamp(); // requests issued
            // ---------------- input: source frame -> dry mix, B-format send, A-format ----------------
            float wet[4] = {0.0F, 0.0F, 0.0F, 0.0F};
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if (c >= channels) continue;
                if (first) {
#pragma unroll
                    for (int o = 0; o < CH; ++o) {
                        const float g = SRC.direct.gains[c][o];
                        if (o < channels && audible(g)) out[o] += in[c] * g;
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float g = SRC.aux[slot].gains[c][k];
                    if (audible(g)) wet[k] += win[c] * g;
                }
            }
            float a[4];
            a[0] = 0.0F; a[0] += wet[0] * b2a; a[0] += wet[1] * b2a; a[0] += wet[2] * b2a; a[0] += wet[3] * b2a;
            a[1] = 0.0F; a[1] += wet[0] * b2a; a[1] += wet[1] * -b2a; a[1] += wet[2] * -b2a; a[1] += wet[3] * b2a;
            a[2] = 0.0F; a[2] += wet[0] * b2a; a[2] += wet[1] * b2a; a[2] += wet[2] * -b2a; a[2] += wet[3] * -b2a;
            a[3] = 0.0F; a[3] += wet[0] * b2a; a[3] += wet[1] * -b2a; a[3] += wet[2] * b2a; a[3] += wet[3] * -b2a;

            gstamp(); // send mix done
            // ---------------- input shelves (reference src/oalsfxpp.cpp:7867-7879) ----------------
            // group 0: a (lp input), group 1: feed-forward sums, group 2: lp output; then hp: 2 -> 1 -> 0
            if (lane < 4) {
                row(0, lane)[3] = lpx0; row(0, lane)[2] = lpx1;
                row(2, lane)[3] = lpy0; row(2, lane)[2] = lpy1; // also the hp input history (hp.x == lp.y)
            }
            if (act) {
#pragma unroll
                for (int c = 0; c < 4; ++c) row(0, c)[4 + lane] = a[c];
            }
            wave_sync();
            if (act) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float* x = row(0, c) + 4 + lane;
                    row(1, c)[4 + lane] = (P.lp.b0 * x[0]) + (P.lp.b1 * x[-1]) + (P.lp.b2 * x[-2]);
                }
            }
            wave_sync();
            if (lane < 4) {
                const float* ra = row(0, lane);
                const float nx1 = ra[4 + L - 2], nx0 = ra[4 + L - 1]; // L == 1: ra[3] is the old newest sample
                lpx1 = nx1; lpx0 = nx0;
                biquad_chain(row(1, lane), row(2, lane), L, lp_a1, lp_a2, lpy0, lpy1);
            }
            wave_sync();
            float xin[4];
            if (eax) {
                if (act) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float* x = row(2, c) + 4 + lane;
                        row(1, c)[4 + lane] = (P.hp.b0 * x[0]) + (P.hp.b1 * x[-1]) + (P.hp.b2 * x[-2]);
                    }
                }
                wave_sync();
                if (lane < 4) {
                    row(0, lane)[3] = hpy0; row(0, lane)[2] = hpy1;
                    biquad_chain(row(1, lane), row(0, lane), L, hp_a1, hp_a2, hpy0, hpy1);
                }
                wave_sync();
#pragma unroll
                for (int c = 0; c < 4; ++c) xin[c] = row(0, c)[4 + lane];
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) xin[c] = row(2, c)[4 + lane];
            }
            if (act) {
#pragma unroll
                for (int c = 0; c < 4; ++c) st(slab_b, r_main.at(c, t4), xin[c]);
            }
            wave_sync();

            gstamp(); // shelves done
            // ---------------- early reflections (reference src/oalsfxpp.cpp:7625-7672) ----------------
            float f[4] = {0.0F, 0.0F, 0.0F, 0.0F};
            for (int sb = 0; sb < L;) {
                const int s = min(L - sb, eap_limit);
                if (act && lane >= sb && lane < sb + s) {
                    float g[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = pre_e ? p_e[j] : tap(slab_b, r_main, j, faded, t4 - 4u * cur_etap[j], t4 - 4u * P.early_tap[j], fade);
                        f[j] = v * P.early_tap_coeff[j];
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float input = f[j];
                        const float d = pre_a ? p_a[j] : tap(slab_b, r_eap, j, faded, t4 - 4u * cur_eap[j], t4 - 4u * P.early_ap_off[j], fade);
                        f[j] = d - (apc * input);
                        g[j] = input + (apc * f[j]);
                    }
                    scatter(g, mx, my);
#pragma unroll
                    for (int j = 0; j < 4; ++j) st(slab_b, r_eap.at(j, t4), g[j]);
                }
                sb += s;
                wave_sync();
            }
            if (act) {
#pragma unroll
                for (int j = 0; j < 4; ++j) st(slab_b, r_eline.at(j, t4), f[3 - j]);
            }
            wave_sync();
            float early[4];
            if (act) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float d = pre_el ? p_el[j] : tap(slab_b, r_eline, j, faded, t4 - 4u * cur_eline[j], t4 - 4u * P.early_line_off[j], fade);
                    f[j] += d * P.early_line_coeff[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) early[j] = f[j];
            {
                float v[4] = {f[3], f[2], f[1], f[0]};
                scatter(v, mx, my);
                if (act) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) st(slab_b, r_main.at(j, t4 - 4u * P.late_feed_tap), v[j]);
                }
            }
            wave_sync();

            gstamp(); // early done
            // ---------------- late reverb (reference src/oalsfxpp.cpp:7735-7794) ----------------
            int md = 0;
            if (mod_active && act) {
                int index = mod_index + done + lane;
                index %= mod_range;
                const float sinus = glibc_sinf(6.28318530717958647692F * index / mod_range);
                md = lround_away(rng[done + lane] * sinus);
            }
            // T60 history prefixes: group 0 = section-1 input, group 2 = section-1 output
            if (lane < 4) {
                row(0, lane)[3] = t60_x;
                row(2, lane)[3] = t60_o1b;
            }
            float late[4] = {0.0F, 0.0F, 0.0F, 0.0F};
            for (int sb = 0; sb < L;) {
                // sub-block: lanes whose late-line feedback reads stay outside the sub-block
                const int rel = lane - sb;
                bool ok = true;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d0 = cur_lline[j] + md;
                    ok &= (d0 > rel) || (d0 <= 0);
                    if (faded) {
                        const int d1 = P.late_line_off[j] + md;
                        ok &= (d1 > rel) || (d1 <= 0);
                    }
                }
                const unsigned long long okm = __ballot(ok || !act || lane < sb) >> sb;
                const int lead = (~okm == 0ULL) ? 64 : __builtin_ctzll(~okm);
                const int s = min(min(L - sb, lap_limit), lead);
                const bool on = act && lane >= sb && lane < sb + s;
                if (on) {
                    const unsigned td4 = t4 - 4u * static_cast<unsigned>(md);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float m = pre_lt ? p_lt[j] : tap(slab_b, r_main, j, faded, t4 - 4u * cur_ltap[j], t4 - 4u * P.late_tap[j], fade);
                        float v = m * P.density_gain;
                        v += pre_ll ? p_ll[j] : tap(slab_b, r_lline, j, faded, td4 - 4u * cur_lline[j], td4 - 4u * P.late_line_off[j], fade);
                        row(0, j)[4 + lane] = v;
                    }
                }
                wave_sync();
                // T60: two first-order sections and the mid gain (reference src/oalsfxpp.cpp:7691-7719);
                // feed-forward halves per lane, feedback halves on the chain lanes
                if (on) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float* x = row(0, j) + 4 + lane;
                        row(1, j)[4 + lane] = (P.t60_lf[j][0] * x[0]) + (P.t60_lf[j][1] * x[-1]);
                    }
                }
                wave_sync();
                if (lane < 4) first_order_chain(row(1, lane), row(2, lane), sb, sb + s, t_l2, 1.0F, false, t60_o1);
                wave_sync();
                if (on) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float* o1 = row(2, j) + 4 + lane;
                        row(1, j)[4 + lane] = (P.t60_hf[j][0] * o1[0]) + (P.t60_hf[j][1] * o1[-1]);
                    }
                }
                wave_sync();
                if (lane < 4) first_order_chain(row(1, lane), row(1, lane), sb, sb + s, t_h2, t_mid, true, t60_o2);
                wave_sync();
                if (on) {
                    float v[4], g[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float input = row(1, j)[4 + lane];
                        const float d = pre_la ? p_la[j] : tap(slab_b, r_lap, j, faded, t4 - 4u * cur_lap[j], t4 - 4u * P.late_ap_off[j], fade);
                        v[j] = d - (apc * input);
                        g[j] = input + (apc * v[j]);
                    }
                    scatter(g, mx, my);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        st(slab_b, r_lap.at(j, t4), g[j]);
                        late[j] = v[j];
                    }
                    float r[4] = {v[3], v[2], v[1], v[0]};
                    scatter(r, mx, my);
#pragma unroll
                    for (int j = 0; j < 4; ++j) st(slab_b, r_lline.at(j, t4), r[j]);
                }
                sb += s;
                wave_sync();
            }
            if (lane < 4) {
                t60_x = row(0, lane)[4 + L - 1];
                t60_o1b = t60_o1; // the second section's last input is the first section's last output
            }

            gstamp(); // late done
            // ---------------- pan to the outputs with gain ramps (reference src/oalsfxpp.cpp:6142-6166) ----------------
            if (ramp_mask != 0ULL) {
                if (g_ramp) {
                    float* gs = gseq + lane * 64;
                    for (int i = 0; i < L; ++i) {
                        gs[i] = g_run;
                        g_run += g_step;
                    }
                }
                wave_sync();
            }
#pragma unroll
            for (int stage = 0; stage < 2; ++stage) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float data = stage ? late[j] : early[j];
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        if (c >= channels) continue;
                        const int q = (stage * 4 + j) * CH + c;
                        if ((ramp_mask >> q) & 1ULL) {
                            out[c] += data * gseq[q * 64 + lane];
                        } else {
                            const float gq = __shfl(g_cur, q);
                            if (audible(gq)) out[c] += data * gq;
                        }
                    }
                }
            }
            if (ramp_mask != 0ULL) wave_sync();

            if (act) {
                if (last) {
                    if (CH == 2) {
                        *reinterpret_cast<float2*>(dst + static_cast<size_t>(pos) * 2) = make_float2(out[0], out[CH - 1]);
                    } else {
#pragma unroll
                        for (int c = 0; c < CH; ++c)
                            if (c < channels) dst[static_cast<size_t>(pos) * channels + c] = out[c];
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < CH; ++c)
                        if (c < channels) mixbuf[c * OALSFX_MAX_CHUNK + pos] = out[c];
                }
            }
        } // tiles

        // ---- chunk epilogue ----
        if (mod_active) mod_filter = rng[todo - 1];
        mod_index = (mod_index + todo) % mod_range;
        offset += todo;
        if (fade_count < OALSFX_RV_FADE_SAMPLES) {
            fade_count += todo;
            if (fade_count >= OALSFX_RV_FADE_SAMPLES) {
                fade_count = OALSFX_RV_FADE_SAMPLES;
                if (lane == 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        utu[ut::TAP4 + 0 + j] = 4u * static_cast<unsigned>(P.early_tap[j]);
                        utu[ut::TAP4 + 4 + j] = 4u * static_cast<unsigned>(P.early_ap_off[j]);
                        utu[ut::TAP4 + 8 + j] = 4u * static_cast<unsigned>(P.early_line_off[j]);
                        utu[ut::TAP4 + 12 + j] = 4u * static_cast<unsigned>(P.late_tap[j]);
                        utu[ut::TAP4 + 16 + j] = 4u * static_cast<unsigned>(P.late_ap_off[j]);
                        utu[ut::TAP4 + 20 + j] = 4u * static_cast<unsigned>(P.late_line_off[j]);
                    }
                }
            }
        }
        if (g_ramp) g_cur = (todo == counter) ? g_tgt : g_run;
        wave_sync();
        base += todo;
    }

    // ---- write the state back ----
    if (lane < 4) {
        S.lp[lane].x[0] = lpx0; S.lp[lane].x[1] = lpx1;
        S.lp[lane].y[0] = lpy0; S.lp[lane].y[1] = lpy1;
        if (eax) {
            S.hp[lane].x[0] = lpy0; S.hp[lane].x[1] = lpy1; // the hp input is the lp output
            S.hp[lane].y[0] = hpy0; S.hp[lane].y[1] = hpy1;
        }
        S.t60[lane][0][0] = t60_x; S.t60[lane][0][1] = t60_o1;
        S.t60[lane][1][0] = t60_o1b; S.t60[lane][1][1] = t60_o2;
    }
    if (q_valid) {
        if (q_stage) S.late_cur_gain[q_line][q_chan] = g_cur;
        else S.early_cur_gain[q_line][q_chan] = g_cur;
    }
    wave_sync();
    if (lane < 24) (&S.cur_early_tap[0])[lane] = static_cast<int32_t>(utu[ut::TAP4 + lane] >> 2);
    if (lane == 0) {
        S.mod_index = mod_index; S.mod_range = mod_range; S.mod_filter = mod_filter;
        S.fade_count = fade_count; S.offset = offset;
        SS.seen_seq = SP.update_seq;
    }
    if (first && !filtered && lane < channels) send_history_follow(ctx, inst, lane, channels, frames, src);
    // settled and at rest (cross-fade over; from which block length on no output gain would be ramped)?  The host reads this back
    // before it lists the instance for a proven-steady build
    if (ctx.exact) {
        const unsigned level = gains_rest_level(q_valid, g_cur, g_tgt);
        if (lane == 0) ctx.exact[sidx] = fade_count >= OALSFX_RV_FADE_SAMPLES ? level : 0u;
    }
}


// Out-of-line entry used by the steady-state kernel for the instances it cannot take: keeps the general path's
// register footprint out of the steady-state tile loop.
template <int CH>
__device__ __noinline__ void reverb_general_call(const KernelCtx* ctx, int slot, int inst, int flags, float* lds, int lane)
{
    reverb_general_instance<CH>(*ctx, slot, inst, flags, lds, lane);
}

template <int CH>
__global__ __launch_bounds__(256, CH == 8 ? 1 : 2) void k_reverb(KernelCtx ctx, int slot, const int* __restrict__ list, int count, int flags)
{
    __shared__ __attribute__((aligned(16))) float lds_all[4][Lds<CH>::kFloats];
    const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + wave_in_block;
    if (w >= count) return; // whole wavefronts leave; the kernel has no workgroup barrier
    const int inst = __builtin_amdgcn_readfirstlane(list[w]);
    reverb_general_instance<CH>(ctx, slot, inst, flags, lds_all[wave_in_block], lane);
}

// A slot that holds ring-light effects for some instances and steady reverbs for others (BASELINE configs[3]): one grid.  The
// first workgroups are groups of the cooperative reverb kernel (its most general build: the reverbs of such a batch rarely
// share properties), the others run one ring-light instance per wavefront.  Two launches on two streams do the same work
// concurrently, but ordering them against the caller's stream costs ~7 us at the fork and ~20 us at the join on this stack.
// FP: every reverb of the grid is proven steady (whole-tile calls): the reverb groups run the build without steady-state test and
// general path, started from hot records.
template <int CH, bool RG, bool FP = false>
__global__ __launch_bounds__(256, 4) void k_slot_mixed(KernelCtx ctx, int slot, const int* __restrict__ steady_list, int steady_count,
                                                       const int* __restrict__ light_list, int light_count, WaveSegments seg, int flags)
{
    union Shared {
        SteadyShared<CH, 4, FP, true, true> steady;
        float light[4][wfx::kLdsFloats];
    };
    __shared__ Shared sh;
    const int steady_groups = (steady_count + 3) >> 2;
    const int group = static_cast<int>(blockIdx.x);
    if (group < steady_groups) {
        // (whole tiles, not proven: the build that keeps instances in transition -- cross-fades, gain ramps -- in the grid)
        reverb_steady_group<CH, 4, false, true, true, true, RG, FP, !RG && !FP>(ctx, slot, steady_list, steady_count, flags, group, sh.steady);
        return;
    }
    // (FP: the build that may be a chained launch -- ctx.turn: its ring-light wavefronts take turns like its reverb groups)
    wfx::wave_block<CH, FP>(ctx, slot, 1, light_list, light_count, seg, flags, group - steady_groups, &sh.light[0][0], wfx::kLdsFloats);
}

void launch_slot_mixed(const KernelCtx& ctx, int slot, const int* steady_list, int steady_count, const int* light_list, int light_count,
                       const WaveSegments& seg, int flags, bool proven, hipStream_t stream, int* groups)
{
    if (groups) *groups = 0;
    if (ctx.frames <= 0 || steady_count + light_count <= 0) return;
    const int light_blocks = seg.n > 0 ? seg.blocks() : (light_count + 3) / 4;
    const dim3 grid((steady_count + 3) / 4 + light_blocks), block(256);
    if (groups) *groups = static_cast<int>(grid.x);
    const bool ragged = (ctx.frames & 63) != 0;
    if (proven && !ragged) {
        if (ctx.channels == 1) OALSFX_LAUNCH((k_slot_mixed<1, false, true>), grid, block, stream, ctx, slot, steady_list, steady_count, light_list, light_count, seg, flags);
        else OALSFX_LAUNCH((k_slot_mixed<2, false, true>), grid, block, stream, ctx, slot, steady_list, steady_count, light_list, light_count, seg, flags);
        return;
    }
    if (ctx.channels == 1 && ragged) OALSFX_LAUNCH((k_slot_mixed<1, true>), grid, block, stream, ctx, slot, steady_list, steady_count, light_list, light_count, seg, flags);
    else if (ctx.channels == 1) OALSFX_LAUNCH((k_slot_mixed<1, false>), grid, block, stream, ctx, slot, steady_list, steady_count, light_list, light_count, seg, flags);
    else if (ragged) OALSFX_LAUNCH((k_slot_mixed<2, true>), grid, block, stream, ctx, slot, steady_list, steady_count, light_list, light_count, seg, flags);
    else OALSFX_LAUNCH((k_slot_mixed<2, false>), grid, block, stream, ctx, slot, steady_list, steady_count, light_list, light_count, seg, flags);
}

// Steady instances: the cooperative tile loop.  Believed steady (`proven` false): an instance that turns out not to be (the kernel
// decides from the device state) falls back to the general path inside the kernel, out of line.  Proven steady: the FP builds.
// carry: some listed instance's write position is off the 128-byte line grid (an odd-sized call came before): the plain FP build that
// holds back every store site's tail (CR == 2) instead of the late feed's only.
// Returns the symbol it launched (every template argument, as rocprofv3 prints them), nullptr when there was nothing to launch.
// template arguments: channels, wavefronts per workgroup, TL, HY, MD, ST, RG, FP, XF, NF, SF, CR
// (a chained step of two kernels -- whole tiles, proven builds only -- takes the variant whose wavefronts allocate 128 registers)
template <int CHv, bool TLv, bool HYv, bool MDv, bool STv, bool RGv, bool FPv, bool XFv, int CRv>
static void launch_steady_build(dim3 grid, dim3 block, hipStream_t stream, const KernelCtx& c, int slot, const int* list, int count, int flags)
{
    if constexpr (FPv && !RGv) {
        if (oalsfx_hip::lds_per_workgroup() > 0) {
            OALSFX_LAUNCH((k_reverb_steady_coop_ep<CHv, 4, TLv, HYv, MDv, STv, RGv, FPv, XFv, false, false, CRv>), grid, block, stream, c, slot, list, count, flags);
            return;
        }
    }
    OALSFX_LAUNCH((k_reverb_steady_coop<CHv, 4, TLv, HYv, MDv, STv, RGv, FPv, XFv, false, false, CRv>), grid, block, stream, c, slot, list, count, flags);
}
#define OALSFX_STEADY(CHv, TLv, HYv, MDv, STv, RGv, FPv, XFv, CRv)                                                                               \
    do {                                                                                                                                         \
        launch_steady_build<CHv, TLv, HYv, MDv, STv, RGv, FPv, XFv, CRv>(grid, block, stream, c, slot, list, count, flags);                       \
        return "k_reverb_steady_coop<" #CHv ", 4, " #TLv ", " #HYv ", " #MDv ", " #STv ", " #RGv ", " #FPv ", " #XFv ", false, false, " #CRv ">";  \
    } while (0)
const char* launch_reverb_steady(const KernelCtx& ctx, int slot, const int* list, int count, int flags, bool close_taps, bool modulated, bool short_taps,
                                 bool proven, bool in_transition, hipStream_t stream, int* groups, bool carry)
{
    if (groups) *groups = count > 0 ? (count + 3) / 4 : 0;
    if (count <= 0) return nullptr;
    const dim3 grid((count + 3) / 4), block(256);
    const KernelCtx& c = ctx;
    const bool ragged = (c.frames & 63) != 0; // the call ends in a partial tile: the most general build's ragged variant
    if (proven && !ragged && c.channels <= 2) {
        // the host has proven every listed instance steady: the builds without steady-state test and general path, started from hot records
        if (c.channels == 1) {
            if (short_taps) OALSFX_STEADY(1, false, true, true, true, false, true, false, 0);
            if (modulated) OALSFX_STEADY(1, false, true, true, false, false, true, false, 0);
            if (close_taps) OALSFX_STEADY(1, false, true, false, false, false, true, false, 0);
            if (carry) OALSFX_STEADY(1, false, false, false, false, false, true, false, 2);
#if OALSFX_CR_FEED
            OALSFX_STEADY(1, false, false, false, false, false, true, false, 1);
#else
            OALSFX_STEADY(1, false, false, false, false, false, true, false, 0);
#endif
        }
        if (short_taps) OALSFX_STEADY(2, false, true, true, true, false, true, false, 0);
        if (modulated) OALSFX_STEADY(2, false, true, true, false, false, true, false, 0);
        if (c.timeline && !close_taps) OALSFX_STEADY(2, true, false, false, false, false, true, false, 0); // (a plain build: no fallback for close taps in an FP launch)
        if (close_taps) OALSFX_STEADY(2, false, true, false, false, false, true, false, 0);
        if (carry) OALSFX_STEADY(2, false, false, false, false, false, true, false, 2);
#if OALSFX_CR_FEED
        OALSFX_STEADY(2, false, false, false, false, false, true, false, 1);
#else
        OALSFX_STEADY(2, false, false, false, false, false, true, false, 0);
#endif
    }
    if (proven && ragged && c.channels <= 2) {
        // a call that ends in a partial tile, every listed instance proven steady and its gains at rest for the call's last block: the
        // ragged variants of the plain and of the most general proven build (hot records, no steady-state test, no general path inside)
        if (c.channels == 1) {
            if (short_taps || modulated || close_taps) OALSFX_STEADY(1, false, true, true, true, true, true, false, 0);
            if (carry) OALSFX_STEADY(1, false, false, false, false, true, true, false, 2);
            OALSFX_STEADY(1, false, false, false, false, true, true, false, 0);
        }
        if (short_taps || modulated || close_taps) OALSFX_STEADY(2, false, true, true, true, true, true, false, 0);
        if (carry) OALSFX_STEADY(2, false, false, false, false, true, true, false, 2);
        OALSFX_STEADY(2, false, false, false, false, true, true, false, 0);
    }
    if (c.channels > 2) {
        // multichannel: the most general build only; the caller launches the general kernel on the same list right after
        if (ragged) OALSFX_STEADY(8, false, true, true, true, true, false, false, 0);
        if (short_taps || modulated) OALSFX_STEADY(8, false, true, true, true, false, false, false, 0);
        if (close_taps) OALSFX_STEADY(8, false, true, false, false, false, false, false, 0);
        OALSFX_STEADY(8, false, false, false, false, false, false, false, 0);
    }
    if (ragged) {
        if (c.channels == 1) OALSFX_STEADY(1, false, true, true, true, true, false, false, 0);
        OALSFX_STEADY(2, false, true, true, true, true, false, false, 0);
    }
    if (in_transition) {
        // some listed instance is folding in a property change (cross-fade, gain ramps): the variant of the most general build that
        // keeps such instances in their workgroup instead of sending them down the general path
        if (c.channels == 1) OALSFX_STEADY(1, false, true, true, true, false, false, true, 0);
        OALSFX_STEADY(2, false, true, true, true, false, false, true, 0);
    }
    if (c.channels == 1) {
        if (short_taps) OALSFX_STEADY(1, false, true, true, true, false, false, false, 0);
        if (modulated) OALSFX_STEADY(1, false, true, true, false, false, false, false, 0);
        if (close_taps) OALSFX_STEADY(1, false, true, false, false, false, false, false, 0);
        OALSFX_STEADY(1, false, false, false, false, false, false, false, 0);
    }
    if (short_taps) OALSFX_STEADY(2, false, true, true, true, false, false, false, 0);
    if (modulated) OALSFX_STEADY(2, false, true, true, false, false, false, false, 0);
    if (c.timeline) OALSFX_STEADY(2, true, false, false, false, false, false, false, 0);
    if (close_taps) OALSFX_STEADY(2, false, true, false, false, false, false, false, 0);
    OALSFX_STEADY(2, false, false, false, false, false, false, false, 0);
}
#undef OALSFX_STEADY

// The steady reverbs of a slot by kind (mono / stereo, whole tiles): counts[0] proven with every tap two tiles away, [1] proven with a
// tap of one to two tiles, [2] proven with shorter taps or a modulated late line, [3] believed steady or in a transition the XF build
// follows; `list` holds them in this order.  One kind alone takes its lean kernel, several share the grid of k_reverb_steady_kinds.
const char* launch_reverb_steady_kinds(const KernelCtx& ctx, int slot, const int* list, const int counts[4], int flags, bool no_fallback, bool filters_inside,
                                       hipStream_t stream, int* groups_out, bool carry)
{
    const int total = counts[0] + counts[1] + counts[2] + counts[3];
    if (groups_out) *groups_out = total > 0 ? (total + 3) / 4 : 0; // (one kind alone: its lean kernel; the grid of kinds says below what it takes)
    if (total <= 0) return nullptr;
    int populated = 0, only = 0;
    for (int k = 0; k < 4; ++k)
        if (counts[k] > 0) { ++populated; only = k; }
    const bool sf = filters_inside && ctx.slots == 1 && counts[0] + counts[1] > 0;
    if (sf) flags |= kFilterInside;
    if (populated == 1 && !(only == 3 && no_fallback)) {
        if (sf) {
            // one of the first two kinds alone, with the send filters inside: template arguments channels, wavefronts, TL, HY, MD, ST, RG, FP, XF, NF, SF
            const dim3 grid((total + 3) / 4), block(256);
            if (ctx.channels == 1 && only == 0) { OALSFX_LAUNCH((k_reverb_steady_coop<1, 4, false, false, false, false, false, true, false, false, true, 0>), grid, block, stream, ctx, slot, list, total, flags); return "k_reverb_steady_coop<1, 4, false, false, false, false, false, true, false, false, true, 0>"; }
            if (ctx.channels == 1) { OALSFX_LAUNCH((k_reverb_steady_coop<1, 4, false, true, false, false, false, true, false, false, true, 0>), grid, block, stream, ctx, slot, list, total, flags); return "k_reverb_steady_coop<1, 4, false, true, false, false, false, true, false, false, true, 0>"; }
            if (only == 0) { OALSFX_LAUNCH((k_reverb_steady_coop<2, 4, false, false, false, false, false, true, false, false, true, 0>), grid, block, stream, ctx, slot, list, total, flags); return "k_reverb_steady_coop<2, 4, false, false, false, false, false, true, false, false, true, 0>"; }
            OALSFX_LAUNCH((k_reverb_steady_coop<2, 4, false, true, false, false, false, true, false, false, true, 0>), grid, block, stream, ctx, slot, list, total, flags);
            return "k_reverb_steady_coop<2, 4, false, true, false, false, false, true, false, false, true, 0>";
        }
        // (close_taps / modulated / short_taps select the FP build of the kind; the believed kind alone: the XF build with the general path inside)
        return launch_reverb_steady(ctx, slot, list, total, flags, only == 1, false, only == 2, only != 3, only == 3, stream, groups_out, carry && only == 0);
    }
    KernelCtx c = ctx;
    c.list_first = -1; // (the kinds read their entries from the list)
    SteadyKinds kinds{};
    int groups = 0;
    for (int k = 0; k < 4; ++k) { kinds.count[k] = counts[k]; groups += kinds.groups(k); }
    if (groups_out) *groups_out = groups;
    const dim3 grid(groups), block(256);
#define OALSFX_KINDS(...)                                                                                  \
    do {                                                                                                   \
        OALSFX_LAUNCH((k_reverb_steady_kinds<__VA_ARGS__>), grid, block, stream, c, slot, list, kinds, flags); \
        return "k_reverb_steady_kinds<" #__VA_ARGS__ ">";                                                  \
    } while (0)
    // template arguments: channels, NF, SF, CR (the plain kind's workgroups; see k_reverb_steady_kinds)
    const bool carry_all = carry && counts[0] > 0 && !sf && !no_fallback;
#if OALSFX_CR_FEED
#define OALSFX_KINDS_BASE(CHv, NFv) OALSFX_KINDS(CHv, NFv, false, 1)
#else
#define OALSFX_KINDS_BASE(CHv, NFv) OALSFX_KINDS(CHv, NFv, false, 0)
#endif
    if (c.channels == 1) {
        if (no_fallback) OALSFX_KINDS_BASE(1, true);
        if (sf) OALSFX_KINDS(1, false, true, 0);
        if (carry_all) OALSFX_KINDS(1, false, false, 2);
        OALSFX_KINDS_BASE(1, false);
    }
    if (no_fallback) OALSFX_KINDS_BASE(2, true);
    if (sf) OALSFX_KINDS(2, false, true, 0);
    if (carry_all) OALSFX_KINDS(2, false, false, 2);
    OALSFX_KINDS_BASE(2, false);
#undef OALSFX_KINDS_BASE
#undef OALSFX_KINDS
}

// Everything else: cross-fades, modulation, gain ramps, taps closer than a tile, partial tiles, more than two channels.
void launch_reverb_general(const KernelCtx& ctx, int slot, const int* list, int count, int flags, hipStream_t stream)
{
    if (count <= 0) return;
    const dim3 grid((count + 3) / 4), block(256);
    const KernelCtx& c = ctx;
    if (c.channels == 1) OALSFX_LAUNCH(k_reverb<1>, grid, block, stream, c, slot, list, count, flags);
    else if (c.channels == 2) OALSFX_LAUNCH(k_reverb<2>, grid, block, stream, c, slot, list, count, flags);
    else OALSFX_LAUNCH(k_reverb<8>, grid, block, stream, c, slot, list, count, flags);
}

} // namespace oalsfx_hip
