// Process kernel for the ring-light effects: null, chorus, flanger, compressor, dedicated, distortion, echo,
// equalizer, ring modulator -- one wavefront per instance, lanes = 64 consecutive sample times (a tile).
//
// Each body replaces the matching EffectState::do_process of the reference (line ranges at each struct).  What is
// parallel over the 64 lanes: the send mix, everything pointwise (wave shapers, carriers, LFOs, panning) and every
// delay-line access whose source lies before the tile.  What stays serial, because it must round exactly like the
// reference: the feedback half of each biquad and the compressor's gain follower run on one "chain" lane per signal
// over an LDS row; delay lines with feedback shorter than a tile are cut into sub-blocks no longer than the delay.
//
// One launch serves every instance of a slot whatever its type (a wave-uniform switch on the descriptor's type), so
// a batch with many effect types mixed (BASELINE configs[3]) still fills the chip with a single grid.
//
// LDS rows are 4 + 64 floats: row[2], row[3] hold the two samples before the tile (the filter history), row[4 + i]
// sample i of the tile; after a tile the last two samples move into the prefix.
#ifndef OALSFX_HIP_WAVE_EFFECTS_BODY_HPP
#define OALSFX_HIP_WAVE_EFFECTS_BODY_HPP

#include "common.hpp"

namespace oalsfx_hip {

// everything of the ring-light effects lives in its own namespace: the header is compiled into wave_effects.hip (their own
// kernel) and into reverb.hip (the kernel that serves a slot holding both ring-light effects and reverbs in one grid)
namespace wfx {

constexpr int kRow = 68;
constexpr int kLdsRows = 20;                  // equalizer: 5 stages x 4 B-format channels
constexpr int kCoefBase = kLdsRows * kRow;    // behind the rows: 4 x 8 floats, the equalizer's band coefficients
constexpr int kLdsFloats = kCoefBase + 32;    // distortion needs 3 x (4 + 256)

typedef __attribute__((address_space(1))) float GlobalFloat;
// Parameters are read through the constant address space: scalar loads, hoisted out of the tile loop.
typedef const __attribute__((address_space(4))) oalsfx_slot_params ConstSlotParams;

struct Coef { float b0, b1, b2, a1, a2; };
template <class B> __device__ __forceinline__ Coef coef(const B& b) { return Coef{b.b0, b.b1, b.b2, b.a1, b.a2}; }

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Feedback half of a biquad over a whole tile: entries [0, N) of a row that holds the feed-forward sums; outputs replace them.
// y = (u - a1*y1) - a2*y2 (reference FilterState::process, src/oalsfxpp.cpp:1009-1014).  row[3], row[2] hold the two outputs
// before the tile.  Every trip count is known to the compiler (chain_biquad below takes the ragged cases).
template <int N>
__device__ __forceinline__ void chain_biquad_whole(float* row, float a1, float a2)
{
    static_assert(N % 32 == 0, "two blocks of sixteen per round");
    float y1 = row[3], y2 = row[2];
    auto step4 = [&](float4& v) {
        v.x = (v.x - (a1 * y1)) - (a2 * y2);
        v.y = (v.y - (a1 * v.x)) - (a2 * y1);
        v.z = (v.z - (a1 * v.y)) - (a2 * v.x);
        v.w = (v.w - (a1 * v.z)) - (a2 * v.y);
        y2 = v.z;
        y1 = v.w;
    };
    // Two register blocks of sixteen samples, A and B, taking turns.  The order of the LDS traffic is chosen for the one place where
    // the compiler waits for all of it, the loop's back edge: what is outstanding there was issued sixteen steps earlier.
    float4* r4 = reinterpret_cast<float4*>(row + 4);
    float4 a0 = r4[0], a1v = r4[1], a2v = r4[2], a3 = r4[3];
    float4 b0 = r4[4], b1 = r4[5], b2 = r4[6], b3 = r4[7];
    step4(a0); step4(a1v); step4(a2v); step4(a3);
    r4[0] = a0; r4[1] = a1v; r4[2] = a2v; r4[3] = a3;
    a0 = r4[8]; a1v = r4[9]; a2v = r4[10]; a3 = r4[11];
    step4(b0); step4(b1); step4(b2); step4(b3);
#pragma unroll 1
    for (int q = 8; q < N / 4; q += 8) {
        r4[q - 4] = b0; r4[q - 3] = b1; r4[q - 2] = b2; r4[q - 1] = b3;
        b0 = r4[q + 4]; b1 = r4[q + 5]; b2 = r4[q + 6]; b3 = r4[q + 7];
        step4(a0); step4(a1v); step4(a2v); step4(a3);
        r4[q + 0] = a0; r4[q + 1] = a1v; r4[q + 2] = a2v; r4[q + 3] = a3;
        // (the last round requests the sixteen floats behind the tile: inside the wavefront's LDS for every caller, never used)
        a0 = r4[q + 8]; a1v = r4[q + 9]; a2v = r4[q + 10]; a3 = r4[q + 11];
        step4(b0); step4(b1); step4(b2); step4(b3);
    }
    r4[N / 4 - 4] = b0; r4[N / 4 - 3] = b1; r4[N / 4 - 2] = b2; r4[N / 4 - 1] = b3;
}

// The same over entries [lo, hi) of the row; row[lo + 3], row[lo + 2] hold the two outputs before `lo`.
__device__ __forceinline__ void chain_biquad(float* row, int lo, int hi, float a1, float a2)
{
    float y1 = row[4 + lo - 1], y2 = row[4 + lo - 2];
    int i = lo;
    auto step4 = [&](float4& v) {
        v.x = (v.x - (a1 * y1)) - (a2 * y2);
        v.y = (v.y - (a1 * v.x)) - (a2 * y1);
        v.z = (v.z - (a1 * v.y)) - (a2 * v.x);
        v.w = (v.w - (a1 * v.z)) - (a2 * v.y);
        y2 = v.z;
        y1 = v.w;
    };
    if ((lo & 3) == 0 && lo + 16 <= hi) {
        // Sixteen samples at a time, entirely in registers, the next sixteen requested before the current ones are worked on:
        // the recurrence then runs at the pace of its instructions (four per sample, 17 cycles on a lone wavefront), with no LDS
        // latency in it.  Two register blocks take turns, so that nothing is moved between them (scripts/micro/chain_step.hip:
        // 34 -> 29 cycles a sample).
        float4* r4 = reinterpret_cast<float4*>(row + 4);
        float4 c0 = r4[(lo >> 2) + 0], c1 = r4[(lo >> 2) + 1], c2 = r4[(lo >> 2) + 2], c3 = r4[(lo >> 2) + 3];
        for (; i + 32 <= hi; i += 32) {
            const int q = i >> 2;
            float4 n0 = r4[q + 4], n1 = r4[q + 5], n2 = r4[q + 6], n3 = r4[q + 7];
            step4(c0); step4(c1); step4(c2); step4(c3);
            r4[q + 0] = c0; r4[q + 1] = c1; r4[q + 2] = c2; r4[q + 3] = c3;
            if (i + 48 <= hi) { c0 = r4[q + 8]; c1 = r4[q + 9]; c2 = r4[q + 10]; c3 = r4[q + 11]; }
            step4(n0); step4(n1); step4(n2); step4(n3);
            r4[q + 4] = n0; r4[q + 5] = n1; r4[q + 6] = n2; r4[q + 7] = n3;
        }
        if (i + 16 <= hi) {
            const int q = i >> 2;
            step4(c0); step4(c1); step4(c2); step4(c3);
            r4[q + 0] = c0; r4[q + 1] = c1; r4[q + 2] = c2; r4[q + 3] = c3;
            i += 16;
        }
    }
    for (; i < hi; ++i) {
        const float y = (row[4 + i] - (a1 * y1)) - (a2 * y2);
        row[4 + i] = y;
        y2 = y1;
        y1 = y;
    }
}

// o[i] = x[i] + c * o[i - 1] over row[4 .. 4 + n) in place, o[-1] = row[3]: sixteen samples at a time in registers, the next sixteen
// requested before the current ones are worked on (two dependent instructions per sample, no LDS latency in the recurrence).
__device__ __forceinline__ void chain_first_order(float* row, int n, float c)
{
    float prev = row[3];
    int i = 0;
    if (n >= 16) {
        float4* r4 = reinterpret_cast<float4*>(row + 4);
        float4 c0 = r4[0], c1 = r4[1], c2 = r4[2], c3 = r4[3];
        auto step4 = [&](float4& v) {
            v.x = v.x + (c * prev);
            v.y = v.y + (c * v.x);
            v.z = v.z + (c * v.y);
            v.w = v.w + (c * v.z);
            prev = v.w;
        };
        for (; i + 32 <= n; i += 32) {   // two register blocks taking turns, as in chain_biquad
            const int q = i >> 2;
            float4 n0 = r4[q + 4], n1 = r4[q + 5], n2 = r4[q + 6], n3 = r4[q + 7];
            step4(c0); step4(c1); step4(c2); step4(c3);
            r4[q + 0] = c0; r4[q + 1] = c1; r4[q + 2] = c2; r4[q + 3] = c3;
            if (i + 48 <= n) { c0 = r4[q + 8]; c1 = r4[q + 9]; c2 = r4[q + 10]; c3 = r4[q + 11]; }
            step4(n0); step4(n1); step4(n2); step4(n3);
            r4[q + 4] = n0; r4[q + 5] = n1; r4[q + 6] = n2; r4[q + 7] = n3;
        }
        if (i + 16 <= n) {
            const int q = i >> 2;
            step4(c0); step4(c1); step4(c2); step4(c3);
            r4[q + 0] = c0; r4[q + 1] = c1; r4[q + 2] = c2; r4[q + 3] = c3;
            i += 16;
        }
    }
    for (; i < n; ++i) {
        prev = row[4 + i] + (c * prev);
        row[4 + i] = prev;
    }
}

// Moves the last two of the n samples a row received into its history prefix.
__device__ __forceinline__ void advance_row(float* row, int n)
{
    if (n >= 2) {
        const float a = row[4 + n - 2], b = row[4 + n - 1];
        row[2] = a; row[3] = b;
    } else if (n == 1) {
        row[2] = row[3]; row[3] = row[4];
    }
}

__device__ __forceinline__ void load_hist(float* xrow, float* yrow, const oalsfx_hist_t& h)
{
    xrow[2] = h.x[1]; xrow[3] = h.x[0];
    yrow[2] = h.y[1]; yrow[3] = h.y[0];
}

__device__ __forceinline__ void store_hist(const float* xrow, const float* yrow, oalsfx_hist_t& h)
{
    h.x[1] = xrow[2]; h.x[0] = xrow[3];
    h.y[1] = yrow[2]; h.y[0] = yrow[3];
}

// Bodies that deliver a tile's output one call late declare `static constexpr int kLag = 1` (see wave_instance).
template <class T, class = void> struct LagOf { static constexpr int value = 0; };
template <class T> struct LagOf<T, decltype(void(T::kLag))> { static constexpr int value = T::kLag; };

// What a body sees of its instance.
struct Inst {
    ConstSlotParams* sp;
    oalsfx_slot_state* ss;
    GlobalFloat* ring;
    float* lds;
    int lane;
    int channels;
    // cooperative workgroups (four instances of one effect type, see chain_phase)
    bool coop;
    int wib;            // this wavefront's place in its workgroup
    int duty;           // the wavefront that runs the workgroup's first chain phase; the duty moves on by one with every phase, so
                        // that the chains of the workgroups that share a CU spread over its four SIMDs whatever their placement
    mutable int phase;  // chain phases so far
    float* group_lds;   // LDS of the workgroup's first wavefront
    int group_stride;   // floats between the LDS areas of consecutive wavefronts
};

// Workgroup barrier that only waits for LDS traffic: global requests (the next tile's inputs) stay in flight across it.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The serial part of a tile: `lines` recurrences per instance, f(lds of the instance, line) runs one of them over rows that
// the parallel part has filled.  A lone wavefront runs its own on its first lanes.  In a cooperative workgroup (four
// wavefronts = four instances of the same effect type, all present) one wavefront runs all 4 * lines on as many lanes while
// the other three wait: a vector instruction costs the same four cycles whether it has one lane or sixteen, and these
// recurrences are most of the instructions of the equalizer, the distortion and the compressor.  Coefficients the
// recurrences need are kept in the rows' unused prefix slots ([0], [1]) so that any wavefront finds them.
template <class F>
__device__ __forceinline__ void chain_phase(const Inst& I, int lines, F&& f)
{
    if (!I.coop) {
        wave_sync();
        if (I.lane < lines) f(I.lds, I.lane);
        wave_sync();
    } else {
        const int duty = (I.duty + I.phase++) & 3;
        lds_barrier();
        if (I.wib == duty && I.lane < 4 * lines) f(I.group_lds + (I.lane / lines) * I.group_stride, I.lane % lines);
        lds_barrier();
    }
}

// std::max / std::min as the reference uses them (first argument wins when the comparison is false, NaNs included)
__device__ __forceinline__ float std_max(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float std_min(float a, float b) { return (b < a) ? b : a; }

template <int CH, class G>
__device__ __forceinline__ void pan(float out[CH], int channels, const G& gains, float v)
{
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const float g = gains[c];
        if (c < channels && audible(g)) out[c] += g * v;
    }
}

// ---------------------------------------------------------------------------------------------
struct NullW {
    __device__ void init(const Inst&) {}
    template <int CH> __device__ void tile(const Inst&, const float*, float*, int) {}
    __device__ void finish(const Inst&) {}
};

// dedicated dialog / LFE (reference src/oalsfxpp.cpp:4556-4576)
struct DedicatedW {
    __device__ void init(const Inst&) {}
    template <int CH> __device__ void tile(const Inst& I, const float* wet, float* out, int)
    {
        const auto& p = I.sp->u.dedicated;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const float g = p.gains[c];
            if (c < I.channels && audible(g)) out[c] += wet[0] * g;
        }
    }
    __device__ void finish(const Inst&) {}
};

#ifndef OALSFX_MODDELAY_AHEAD
#define OALSFX_MODDELAY_AHEAD 1 // 0: A/B builds without the request one tile ahead
#endif
// chorus / flanger (reference src/oalsfxpp.cpp:4113-4276, 5384-5547): buf[o] = in; t = buf[o - d] * feedback; buf[o] += t; out = t
struct ModDelayW {
    // (The LFO's phase, `offset % lfo_range`, is an integer division by a run-time divisor per lane, twice per tile.  Round 4 carried
    // the remainder from tile to tile by additions instead -- one division per call -- and measured it: 4096 choruses 12.6 -> 13.7 us
    // per buffer, flangers 12.6 -> 13.7, config 3 93.2 -> 96.8, with the branch on a wave-uniform flag or not
    // (profiles/r04c_instruction_diet/lfo_phase_by_additions.txt).  The tile is a chain of latencies behind the dependent ring load;
    // the division was never on it.  Taken out again.)
    int offset;
    // Round 4: the ring values of the next tile, requested one tile ahead when every lane's delay is at least two tiles long (the chorus:
    // 16 ms +- its depth; what the next tile reads then lies before this tile's writes) -- the tile is the latency of that dependent load
    // and little else, so the next tile's is put behind this tile's work, as the echo's taps are.
    float v_next[2];
    bool have_next, may_ask; // may_ask: the LFO's lowest delay is two tiles long (a flanger's is not: it would pay for the second delay per tile and never ask)
    __device__ void init(const Inst& I)
    {
        const auto& p = I.sp->u.moddelay;
        offset = I.ss->u.moddelay.offset;
        may_ask = p.delay - static_cast<int>(fabsf(p.depth)) - 1 >= 128;
        have_next = false;
        v_next[0] = v_next[1] = 0.0F;
    }
    template <class P> __device__ static int lfo_delay(const P& p, int phase)
    {
        if (p.waveform == 1) return static_cast<int>((1.0F - fabsf(2.0F - (p.lfo_scale * phase))) * p.depth) + p.delay;
        return static_cast<int>(glibc_sinf(p.lfo_scale * phase) * p.depth) + p.delay;
    }
    template <int CH> __device__ void tile(const Inst& I, const float* wet, float* out, int L)
    {
        const auto& p = I.sp->u.moddelay;
        const int lane = I.lane;
        const unsigned mask = static_cast<unsigned>(p.ring_len - 1);
        const int o = offset + lane;
        const float in = wet[0];
        const float fb = p.feedback;
        float t[2];
        // Sample i reads what sample i - d wrote (d == 0: its own input).  Sources before the tile come from the ring in one
        // round of loads for both sides; sources inside the tile are handed from lane to lane, the lanes whose source is
        // settled going together.  The tile's 64 new ring values per side are stored once, at the end.
        int d[2];
        bool inside[2];
        float v[2];
        if (have_next) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                d[k] = 128;        // (at least: no source inside the tile, and nothing else asks)
                inside[k] = false;
                v[k] = lane < L ? v_next[k] : in;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const GlobalFloat* buf = I.ring + (k ? p.ring_len : 0);
                const int phase = (k ? o + p.lfo_disp : o) % p.lfo_range;
                d[k] = lfo_delay(p, phase);
                inside[k] = d[k] > 0 && lane - d[k] >= 0; // written by an earlier lane of this tile
                v[k] = in;
                if (lane < L && d[k] != 0 && !inside[k]) v[k] = buf[static_cast<unsigned>(o - d[k]) & mask];
            }
        }
        // the next tile's sources, while this tile is worked on (a call's last tile asks for nothing: L < 64, or one request too many)
        have_next = false;
        if (OALSFX_MODDELAY_AHEAD && may_ask && L == 64) {
            int dn[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) dn[k] = lfo_delay(p, (k ? o + 64 + p.lfo_disp : o + 64) % p.lfo_range);
            have_next = __ballot(dn[0] < 128 || dn[1] < 128) == 0ULL;
            if (have_next) {
#pragma unroll
                for (int k = 0; k < 2; ++k) v_next[k] = (I.ring + (k ? p.ring_len : 0))[static_cast<unsigned>(o + 64 - dn[k]) & mask];
            }
        }
        // both sides walk the tile together: a side with a short delay needs many rounds (64 / delay), each one a ballot and a
        // lane permute whose latencies the other side's round hides
        float val[2] = {0.0F, 0.0F}; // what this lane's sample leaves in the two rings
        t[0] = t[1] = 0.0F;
        int s[2] = {0, 0};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            // A delay of a few samples would take 64 / delay such rounds.  Where every in-tile source of a side lies the same d
            // samples back (the LFO moves slowly: the usual case), sample i depends on i - d, i - 2d, ...: d independent chains,
            // which d lanes walk through an LDS row, each a serial multiply-add per step -- 64 / d steps of two dependent
            // instructions instead of 64 / d rounds of ballot and permute.
            const unsigned long long in_tile = __ballot(inside[k] && lane < L);
            if (in_tile == 0ULL) continue;
            const int d0 = __builtin_amdgcn_readlane(d[k], static_cast<int>(__builtin_ctzll(in_tile)));
            if (d0 > 16 || __ballot(lane >= d0 && lane < L && d[k] != d0) != 0ULL) continue;
            // chain c = i % d0 gets a row of its own: its head (the one sample that reads the ring) in the row's prefix slot [3], the
            // samples behind it in order from [4] on, so that the chain lane walks contiguous floats in register blocks
            const int chain = lane % d0, step = lane / d0; // step 0: head
            float* row = I.lds + chain * kRow;
            const float head_t = v[k] * fb;
            if (lane < L) row[3 + step] = step == 0 ? in + head_t : in;
            wave_sync();
            if (lane < d0 && lane < L) chain_first_order(I.lds + lane * kRow, (L - lane + d0 - 1) / d0 - 1, fb);
            wave_sync();
            if (lane < L) {
                val[k] = row[3 + step];
                t[k] = step == 0 ? head_t : row[2 + step] * fb; // what the sample d0 before left in the ring, fed back
            }
            wave_sync();
            s[k] = L;
        }
        while (s[0] < L || s[1] < L) {
            int e[2];
            float handed[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const bool blocked = lane >= s[k] && lane < L && inside[k] && lane - d[k] >= s[k];
                const unsigned long long nb = __ballot(blocked);
                e[k] = nb ? static_cast<int>(__builtin_ctzll(nb)) : L;
                handed[k] = __shfl(val[k], inside[k] ? lane - d[k] : lane);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (lane >= s[k] && lane < L && lane < e[k]) {
                    if (inside[k]) v[k] = handed[k];
                    t[k] = v[k] * fb;
                    val[k] = in + t[k];
                }
                s[k] = max(s[k], e[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            GlobalFloat* buf = I.ring + (k ? p.ring_len : 0);
            if (lane < L) buf[static_cast<unsigned>(o) & mask] = val[k];
        }
        // left tap before right tap for every output (reference :4193-4208)
#pragma unroll
        for (int k = 0; k < 2; ++k) pan<CH>(out, I.channels, p.gains[k], t[k]);
        // `out += t * g` in the reference; the product commutes
        offset += L;
        wave_sync();
    }
    __device__ void finish(const Inst& I)
    {
        if (I.lane == 0) I.ss->u.moddelay.offset = offset;
    }
};

// compressor (reference src/oalsfxpp.cpp:4352-4453)
struct CompressorW {
    __device__ void init(const Inst& I)
    {
        if (I.lane == 0) {
            I.lds[3] = I.ss->u.compressor.gain_control;
            I.lds[0] = I.sp->u.compressor.attack_rate;
            I.lds[1] = I.sp->u.compressor.release_rate;
        }
        wave_sync();
    }
    template <int CH> __device__ void tile(const Inst& I, const float* wet, float* out, int L)
    {
        const auto& p = I.sp->u.compressor;
        float* row = I.lds;
        float amplitude = 1.0F;
        if (p.enabled) {
            amplitude = fabsf(wet[0]);
            amplitude = std_max(amplitude + fabsf(wet[1]), std_max(amplitude + fabsf(wet[2]), amplitude + fabsf(wet[3])));
        }
        row[4 + I.lane] = amplitude;
        chain_phase(I, 1, [L](float* row, int) {
            // the gain follower is a serial min/max recurrence (reference :4385-4400)
            float gc = row[3];
            const float attack = row[0], release = row[1];
            auto follow = [&](float a) {
                if (a > gc) gc = std_min(gc + attack, a);
                else if (a < gc) gc = std_max(gc - release, a);
                return gc;
            };
            int i = 0;
            if (L >= 4) {
                float4 a = *reinterpret_cast<const float4*>(row + 4); // one request ahead of the dependent arithmetic
                for (; i + 4 <= L; i += 4) {
                    float4 an = a;
                    if (i + 8 <= L) an = *reinterpret_cast<const float4*>(row + 8 + i);
                    float4 g;
                    g.x = follow(a.x); g.y = follow(a.y); g.z = follow(a.z); g.w = follow(a.w);
                    *reinterpret_cast<float4*>(row + 4 + i) = g;
                    a = an;
                }
            }
            for (; i < L; ++i) row[4 + i] = follow(row[4 + i]);
            row[3] = gc;
        });
        const float output = 1.0F / std_min(2.0F, std_max(0.5F, row[4 + I.lane])); // Math::clamp(gc, 0.5, 2) = min(max_value, max(min_value, gc))
        wave_sync();
#pragma unroll
        for (int j = 0; j < 4; ++j) pan<CH>(out, I.channels, p.gains[j], wet[j] * output);
    }
    __device__ void finish(const Inst& I)
    {
        if (I.lane == 0) I.ss->u.compressor.gain_control = I.lds[3];
    }
};

// equalizer (reference src/oalsfxpp.cpp:5161-5213): four cascaded biquads on each B-format channel.
// Row (stage * 4 + channel): stage 0 the input, stage b + 1 the output of band b; a row's prefix holds the two samples
// before the tile, which are both the band's output history and the next band's input history.
// The sixteen filters of an instance run as a pipeline on sixteen lanes: lane (band b, channel c) filters 16-sample block
// r - b in round r, one block behind the band that feeds it, so a tile takes 4 + 3 rounds of 16 serial steps instead of
// 4 x 64.  Each lane computes its filter whole (feed-forward sums and recurrence, (b0 x0 + b1 x1) + b2 x2 then
// (u - a1 y1) - a2 y2 as FilterState::process rounds them).
struct EqualizerW {
    static constexpr int kBlock = 16;
    __device__ void init(const Inst& I)
    {
        const oalsfx_equalizer_state& s = I.ss->u.equalizer;
        if (I.lane < 16) {
            const int b = I.lane >> 2, ch = I.lane & 3;
            load_hist(I.lds + (b * 4 + ch) * kRow, I.lds + ((b + 1) * 4 + ch) * kRow, s.hist[b][ch]);
        }
        if (I.lane < 4) {
            const auto& band = I.sp->u.equalizer.band[I.lane];
            float* cf = I.lds + kCoefBase + 8 * I.lane; // where any wavefront of a cooperative workgroup finds them
            cf[0] = band.b0; cf[1] = band.b1; cf[2] = band.b2; cf[3] = band.a1; cf[4] = band.a2;
        }
        wave_sync();
    }
    // one lane, one filter, samples [n0, n1) of its rows
    __device__ static void filter_block(const float* xrow, float* yrow, int n0, int n1, const float* cf)
    {
        const float b0 = cf[0], b1 = cf[1], b2 = cf[2], a1 = cf[3], a2 = cf[4];
        float x1 = xrow[4 + n0 - 1], x2 = xrow[4 + n0 - 2], y1 = yrow[4 + n0 - 1], y2 = yrow[4 + n0 - 2];
        auto step = [&](float x0) {
            const float u = ((b0 * x0) + (b1 * x1)) + (b2 * x2);
            const float y = (u - (a1 * y1)) - (a2 * y2);
            x2 = x1; x1 = x0; y2 = y1; y1 = y;
            return y;
        };
        if (n1 - n0 == kBlock) {
            const float4* x4 = reinterpret_cast<const float4*>(xrow + 4 + n0);
            float4* y4 = reinterpret_cast<float4*>(yrow + 4 + n0);
            float4 v[4] = {x4[0], x4[1], x4[2], x4[3]};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k].x = step(v[k].x); v[k].y = step(v[k].y); v[k].z = step(v[k].z); v[k].w = step(v[k].w);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) y4[k] = v[k];
        } else {
            for (int n = n0; n < n1; ++n) yrow[4 + n] = step(xrow[4 + n]);
        }
    }
    template <int CH> __device__ void tile(const Inst& I, const float* wet, float* out, int L)
    {
        const auto& p = I.sp->u.equalizer;
        const int lane = I.lane;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) I.lds[ch * kRow + 4 + lane] = wet[ch];
        chain_phase(I, 16, [L](float* lds, int line) {
            const int b = line >> 2, ch = line & 3;
            const float* xrow = lds + (b * 4 + ch) * kRow;
            float* yrow = lds + ((b + 1) * 4 + ch) * kRow;
            const float* cf = lds + kCoefBase + 8 * b;
            const int blocks = (L + kBlock - 1) / kBlock;
            for (int r = 0; r < blocks + 3; ++r) {
                const int k = r - b;
                if (k >= 0 && k < blocks) filter_block(xrow, yrow, k * kBlock, min(L, (k + 1) * kBlock), cf);
                wave_sync(); // the band behind reads this block in the next round
            }
        });
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) pan<CH>(out, I.channels, p.gains[ch], I.lds[(16 + ch) * kRow + 4 + lane]);
        wave_sync();
        if (lane < 20) advance_row(I.lds + lane * kRow, L);
        wave_sync();
    }
    __device__ void finish(const Inst& I)
    {
        oalsfx_equalizer_state& s = I.ss->u.equalizer;
        if (I.lane < 16) {
            const int b = I.lane >> 2, ch = I.lane & 3;
            store_hist(I.lds + (b * 4 + ch) * kRow, I.lds + ((b + 1) * 4 + ch) * kRow, s.hist[b][ch]);
        }
    }
};

// ring modulator (reference src/oalsfxpp.cpp:5652-5784): one-pole high-pass per B-format channel, times the carrier.
// Rows 0..3 input, 4..7 filter output.
struct RingModW {
    int index;
    __device__ void init(const Inst& I)
    {
        const oalsfx_ringmod_state& s = I.ss->u.ringmod;
        index = s.index;
        if (I.lane < 4) {
            load_hist(I.lds + I.lane * kRow, I.lds + (4 + I.lane) * kRow, s.hist[I.lane]);
            float* yrow = I.lds + (4 + I.lane) * kRow;
            yrow[0] = I.sp->u.ringmod.filter.a1;
            yrow[1] = I.sp->u.ringmod.filter.a2;
        }
        wave_sync();
    }
    __device__ static float carrier(int waveform, int idx)
    {
        constexpr int frac_bits = 24;
        constexpr int frac_one = 1 << frac_bits;
        if (waveform == 0) return glibc_sinf(idx * (6.28318530717958647692F / frac_one) - 3.14159265358979323846F) * 0.5F + 0.5F;
        if (waveform == 1) return static_cast<float>(idx) / frac_one;
        return static_cast<float>((idx >> (frac_bits - 1)) & 1);
    }
    template <int CH> __device__ void tile(const Inst& I, const float* wet, float* out, int L)
    {
        const auto& p = I.sp->u.ringmod;
        const int lane = I.lane;
        const Coef c = coef(p.filter);
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) I.lds[ch * kRow + 4 + lane] = wet[ch];
        wave_sync();
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const float* x = I.lds + ch * kRow + 4 + lane;
            I.lds[(4 + ch) * kRow + 4 + lane] = ((c.b0 * x[0]) + (c.b1 * x[-1])) + (c.b2 * x[-2]);
        }
        chain_phase(I, 4, [L](float* lds, int line) {
            float* row = lds + (4 + line) * kRow;
            chain_biquad(row, 0, L, row[0], row[1]);
        });
        // the carrier index is pre-incremented: sample i sees index + (i + 1) * step (reference :5748-5752)
        const unsigned frac_mask = (1u << 24) - 1u;
        const int idx = static_cast<int>((static_cast<unsigned>(index) + static_cast<unsigned>(lane + 1) * static_cast<unsigned>(p.step)) & frac_mask);
        const float m = carrier(p.waveform, idx);
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) pan<CH>(out, I.channels, p.gains[ch], I.lds[(4 + ch) * kRow + 4 + lane] * m);
        index = static_cast<int>((static_cast<unsigned>(index) + static_cast<unsigned>(L) * static_cast<unsigned>(p.step)) & frac_mask);
        wave_sync();
        if (lane < 8) advance_row(I.lds + lane * kRow, L);
        wave_sync();
    }
    __device__ void finish(const Inst& I)
    {
        oalsfx_ringmod_state& s = I.ss->u.ringmod;
        if (I.lane < 4) store_hist(I.lds + I.lane * kRow, I.lds + (4 + I.lane) * kRow, s.hist[I.lane]);
        if (I.lane == 0) s.index = index;
    }
};

// echo (reference src/oalsfxpp.cpp:4887-4962): t1 = ring[o - tap1], t2 = ring[o - tap2]; in = t2 + x; ring[o] = biquad(in) * feed.
// Row 0 the filter input, row 1 its output.  tap1 <= tap2, so sub-blocks of min(64, tap1) samples read settled data only.
struct EchoW {
    int offset;
    float n_t1, n_t2;   // taps of the next tile, requested one tile ahead when both taps are at least two tiles long
    bool have_next;
    int block;          // samples per sub-block: no longer than the tap the feedback runs through, tap2 (in a cooperative workgroup: than
                        // anybody's); tap1 only listens, what it hears inside the tile is read when the tile is complete
    __device__ void init(const Inst& I)
    {
        const auto& p = I.sp->u.echo;
        offset = I.ss->u.echo.offset;
        n_t1 = n_t2 = 0.0F;
        have_next = false;
        block = max(1, min(64, p.tap2 > 0 ? p.tap2 : 64));
        if (I.lane == 0) {
            load_hist(I.lds, I.lds + kRow, I.ss->u.echo.filter);
            float* yrow = I.lds + kRow; // the recurrence's coefficients travel with its row
            yrow[0] = p.filter.a1; yrow[1] = p.filter.a2;
            if (I.coop) reinterpret_cast<int*>(I.lds + kCoefBase)[0] = block;
        }
        wave_sync();
        if (I.coop) {
            // the four instances walk the tile in the same sub-blocks, so that their recurrences can share a wavefront
            lds_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) block = min(block, reinterpret_cast<const int*>(I.group_lds + k * I.group_stride + kCoefBase)[0]);
        }
    }
    template <int CH> __device__ void tile(const Inst& I, const float* wet, float* out, int L)
    {
        const auto& p = I.sp->u.echo;
        const int lane = I.lane;
        const unsigned mask = static_cast<unsigned>(p.ring_len - 1);
        const Coef c = coef(p.filter);
        const int o = offset + lane;
        float* xrow = I.lds;
        float* yrow = I.lds + kRow;
        float* wrow = I.lds + 2 * kRow; // what this tile writes to the ring; taps shorter than the tile read it here
        // taps whose source lies before the tile: one round of ring loads
        const bool in1 = p.tap1 > 0 && lane - p.tap1 >= 0, in2 = p.tap2 > 0 && lane - p.tap2 >= 0;
        float t1 = 0.0F, t2 = 0.0F;
        if (have_next) {
            t1 = n_t1; t2 = n_t2;
        } else if (lane < L) {
            if (!in1) t1 = I.ring[static_cast<unsigned>(o - p.tap1) & mask];
            if (!in2) t2 = I.ring[static_cast<unsigned>(o - p.tap2) & mask];
        }
        // the next tile's taps lie before this tile's writes when both are at least two tiles long: request them now
        have_next = L == 64 && p.tap1 >= 128 && p.tap2 >= 128;
        if (have_next) {
            n_t1 = I.ring[static_cast<unsigned>(o + 64 - p.tap1) & mask];
            n_t2 = I.ring[static_cast<unsigned>(o + 64 - p.tap2) & mask];
        }
        for (int s = 0; s < L; s += block) {
            const int e = min(L, s + block);
            const bool mine = lane >= s && lane < e;
            if (mine) {
                if (in2) t2 = wrow[4 + lane - p.tap2];
                xrow[4 + lane] = t2 + wet[0];
            }
            wave_sync();
            if (mine) {
                const float* x = xrow + 4 + lane;
                yrow[4 + lane] = ((x[0] * c.b0) + (x[-1] * c.b1)) + (x[-2] * c.b2);
            }
            chain_phase(I, 1, [s, e](float* lds, int) {
                float* row = lds + kRow;
                chain_biquad(row, s, e, row[0], row[1]);
            });
            if (mine) wrow[4 + lane] = yrow[4 + lane] * p.feed_gain;
            wave_sync();
        }
        if (lane < L) I.ring[static_cast<unsigned>(o) & mask] = wrow[4 + lane];
        if (in1 && lane < L) t1 = wrow[4 + lane - p.tap1];
#pragma unroll
        for (int ch = 0; ch < CH; ++ch) {
            if (ch >= I.channels) continue;
            const float g0 = p.gains[0][ch];
            if (audible(g0)) out[ch] += t1 * g0;
            const float g1 = p.gains[1][ch];
            if (audible(g1)) out[ch] += t2 * g1;
        }
        offset += L;
        wave_sync();
        if (lane < 2) advance_row(I.lds + lane * kRow, L);
        wave_sync();
    }
    __device__ void finish(const Inst& I)
    {
        if (I.lane == 0) {
            store_hist(I.lds, I.lds + kRow, I.ss->u.echo.filter);
            I.ss->u.echo.offset = offset;
        }
    }
};

// distortion (reference src/oalsfxpp.cpp:4675-4750): 4x zero-stuffed oversampling, low-pass, three-stage wave shaper,
// band-pass, keep every fourth sample.  Arrays of 4 + 256 floats: low-pass sums/outputs, (the shaped samples' two history slots),
// band-pass sums/outputs; oversampled sample n of the tile belongs to frame n / 4.
struct DistortionW {
    static constexpr int kArr = 4 + 4 * 64;
    float x_hist0, x_hist1; // low-pass input history (zeros after the first frame: the stuffed samples)
    __device__ void init(const Inst& I)
    {
        const oalsfx_distortion_state& s = I.ss->u.distortion;
        x_hist0 = s.low_pass.x[0];
        x_hist1 = s.low_pass.x[1];
        if (I.lane == 0) {
            float* lp = I.lds; float* sh = I.lds + kArr; float* bp = I.lds + 2 * kArr;
            lp[2] = s.low_pass.y[1]; lp[3] = s.low_pass.y[0];
            sh[2] = s.band_pass.x[1]; sh[3] = s.band_pass.x[0];
            bp[2] = s.band_pass.y[1]; bp[3] = s.band_pass.y[0];
            lp[0] = I.sp->u.distortion.low_pass.a1; lp[1] = I.sp->u.distortion.low_pass.a2;
            bp[0] = I.sp->u.distortion.band_pass.a1; bp[1] = I.sp->u.distortion.band_pass.a2;
        }
        wave_sync();
    }
    // The body lags by one tile (kLag): a call takes the B-format send of tile k (L frames, none in the flushing call behind the
    // last tile) and adds to `out` the output of tile k - 1 (Lp frames, none in the first call).  The low-pass chain of tile k
    // and the band-pass chain of tile k - 1 do not depend on each other: they run in the same chain phase on two lanes per
    // instance, so a tile costs one 256-step recurrence instead of two in a row.
    static constexpr int kLag = 1;
    template <int CH> __device__ void tile(const Inst& I, const float* wet, float* out, int L, int Lp)
    {
        const auto& p = I.sp->u.distortion;
        const int lane = I.lane;
        float* lp = I.lds; float* sh = I.lds + kArr; float* bp = I.lds + 2 * kArr;
        const int n = 4 * L, np = 4 * Lp;
        if (L > 0) {
            // low-pass feed-forward sums of this frame's four oversampled inputs (X, 0, 0, 0)
            const Coef c = coef(p.low_pass);
            const float X = wet[0] * 4.0F;
            const float z = 0.0F;
            const float p1 = lane == 0 ? x_hist0 : z; // input before this frame
            const float p2 = lane == 0 ? x_hist1 : z;
            float4 u;
            u.x = ((c.b0 * X) + (c.b1 * p1)) + (c.b2 * p2);
            u.y = ((c.b0 * z) + (c.b1 * X)) + (c.b2 * p1);
            u.z = ((c.b0 * z) + (c.b1 * z)) + (c.b2 * X);
            u.w = ((c.b0 * z) + (c.b1 * z)) + (c.b2 * z);
            *reinterpret_cast<float4*>(lp + 4 + 4 * lane) = u;
        }
        // line 0: the low-pass recurrence of this tile; line 1: the band-pass recurrence of the tile before (whose feed-forward
        // sums the call before left in its row); the same code on two lanes
        chain_phase(I, 2, [n, np](float* lds, int line) {
            float* row = lds + (line ? 2 * kArr : 0);
            const int cnt = line ? np : n;
            if (cnt == 256) chain_biquad_whole<256>(row, row[0], row[1]);
            else chain_biquad(row, 0, cnt, row[0], row[1]);
        });
        if (Lp > 0) {
            const float kept = bp[4 + 4 * lane];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const float g = p.gains[c] * p.attenuation;
                if (c < I.channels && audible(g)) out[c] += g * kept;
            }
        }
        wave_sync();
        if (lane == 0) advance_row(bp, np); // the band-pass outputs become history before the row takes this tile's sums
        if (L > 0) {
            // wave shaper and band-pass feed-forward sums, four consecutive oversampled samples per lane, in registers: the two
            // shaped samples before a lane's four come from the lane below (from the row's history slots for lane 0)
            const float h2 = sh[2], h1 = sh[3];
            const float4 y = *reinterpret_cast<const float4*>(lp + 4 + 4 * lane);
            const float fc = p.edge_coeff;
            auto shape = [fc](float smp) {
                smp = (1.0F + fc) * smp / (1.0F + (fc * fabsf(smp)));
                smp = (1.0F + fc) * smp / (1.0F + (fc * fabsf(smp))) * -1.0F;
                smp = (1.0F + fc) * smp / (1.0F + (fc * fabsf(smp)));
                return smp;
            };
            const float x0 = shape(y.x), x1 = shape(y.y), x2 = shape(y.z), x3 = shape(y.w);
            float p1 = __shfl_up(x3, 1), p2 = __shfl_up(x2, 1);
            if (lane == 0) { p1 = h1; p2 = h2; }
            const Coef c = coef(p.band_pass);
            float4 u;
            u.x = ((c.b0 * x0) + (c.b1 * p1)) + (c.b2 * p2);
            u.y = ((c.b0 * x1) + (c.b1 * x0)) + (c.b2 * p1);
            u.z = ((c.b0 * x2) + (c.b1 * x1)) + (c.b2 * x0);
            u.w = ((c.b0 * x3) + (c.b1 * x2)) + (c.b2 * x1);
            *reinterpret_cast<float4*>(bp + 4 + 4 * lane) = u;
            if (lane == L - 1) { sh[2] = x2; sh[3] = x3; } // the shaped samples' history for the next tile
            if (lane == 0) advance_row(lp, n);
            x_hist0 = 0.0F; x_hist1 = 0.0F;
            wave_sync();
        }
    }
    __device__ void finish(const Inst& I)
    {
        if (I.lane == 0) {
            oalsfx_distortion_state& s = I.ss->u.distortion;
            const float* lp = I.lds; const float* sh = I.lds + kArr; const float* bp = I.lds + 2 * kArr;
            s.low_pass.x[0] = x_hist0; s.low_pass.x[1] = x_hist1;
            s.low_pass.y[1] = lp[2]; s.low_pass.y[0] = lp[3];
            s.band_pass.x[1] = sh[2]; s.band_pass.x[0] = sh[3];
            s.band_pass.y[1] = bp[2]; s.band_pass.y[0] = bp[3];
        }
    }
};

static_assert(3 * DistortionW::kArr <= kLdsFloats, "distortion arrays must fit the per-wave LDS");

// One instance on one wavefront: the front end (dry mix / B-format send of mix_source, reference
// src/oalsfxpp.cpp:2917-2982), the effect body tile by tile, the back end (write_f32, :3414-3431).
// How a wavefront's workgroup is made up: `coop` says its four wavefronts hold four instances of one effect type and run
// their recurrences together (chain_phase).
struct Group {
    bool coop;
    int wib, duty;
    float* lds;  // of the workgroup's first wavefront
    int stride;  // floats between the wavefronts' LDS areas
};

template <int CH, class Fx>
__device__ __forceinline__ void wave_instance(const KernelCtx& ctx, int slot, int inst, int flags, float* lds, int lane, const Group& group)
{
    const int channels = (CH == 8) ? ctx.channels : CH;
    const int frames = ctx.frames;
    const size_t sidx = static_cast<size_t>(inst) * ctx.slots + slot;
    // send gains through the constant address space: scalar loads, hoisted out of the tile loop
    typedef const __attribute__((address_space(4))) oalsfx_source_params ConstSourceParams;
    ConstSourceParams& SRC = *(ConstSourceParams*)(uintptr_t)(ctx.source + inst);
    const bool first = (flags & kFirst) != 0;
    const bool last = (flags & kLast) != 0;
    const bool filtered = (flags & kFiltered) != 0 && instance_has_send_filter(ctx, inst);
    const float* src = ctx.raw_src + static_cast<size_t>(inst) * ctx.io_stride;
    const float* wsrc = src;
    if (filtered) {
        src = ctx.src + static_cast<size_t>(inst) * ctx.src_stride;
        wsrc = ctx.wet_src + static_cast<size_t>(inst) * ctx.src_stride;
    }
    float* dst = ctx.dst + static_cast<size_t>(inst) * ctx.io_stride;
    float* mixbuf = ctx.mixbuf ? ctx.mixbuf + static_cast<size_t>(inst) * channels * OALSFX_MAX_CHUNK : nullptr;
    const bool send_on = SRC.aux[slot].out_channels != 0;

    Inst I;
    I.sp = (ConstSlotParams*)(uintptr_t)(ctx.params + sidx);
    I.ss = ctx.state + sidx;
    I.ring = (GlobalFloat*)(uintptr_t)ctx.rings[sidx];
    I.lds = lds;
    I.lane = lane;
    I.channels = channels;
    I.coop = group.coop;
    I.wib = group.wib;
    I.duty = group.duty;
    I.phase = 0;
    I.group_lds = group.lds;
    I.group_stride = group.stride;

    // measurement only (OALSFX_DEBUG_TIMELINE): every 64th instance stamps the shader clock around the parts of its tiles, 24 stamps per
    // slot (scripts/timeline_wave.py, timeline_slots.py)
    int ts_i = 0;
    auto stamp = [&]() {
        if (ctx.timeline && (inst & 63) == 0 && (inst >> 6) < 64 && lane == 0 && ts_i < 24)
            ctx.timeline[64 * 4 * 96 + (inst >> 6) * 96 + (slot & 3) * 24 + ts_i++] = clock64();
    };
    stamp();
    Fx fx;

    // the inputs of a tile (source frame, filtered send input, accumulated mix of the earlier slots) do not depend on the effect:
    // they are requested one tile ahead, so that their latency hides behind the body of the current tile
    float n_in[CH], n_win[CH], n_mix[CH];
    auto request = [&](int base) {
        const int p = base + lane;
        const bool a = p < frames;
#pragma unroll
        for (int c = 0; c < CH; ++c) { n_in[c] = 0.0F; n_mix[c] = 0.0F; }
        if (a) {
            if (CH == 2) {
                const float2 v = *reinterpret_cast<const float2*>(src + static_cast<size_t>(p) * 2);
                n_in[0] = v.x; n_in[CH - 1] = v.y;
            } else {
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    if (c < channels) n_in[c] = src[static_cast<size_t>(p) * channels + c];
            }
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) n_win[c] = n_in[c];
        if (filtered && a) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (c < channels) n_win[c] = wsrc[static_cast<size_t>(p) * channels + c];
        }
        if (!first && a) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (c < channels) n_mix[c] = mixbuf[c * OALSFX_MAX_CHUNK + p];
        }
    };
    auto store_tile = [&](const float* o, int p, bool a) {
        if (!a) return;
        if (last && ctx.turn_set != 0u && CH <= 2) {
            // a chained launch (the mixed grid; ring-light batches under the test switch): the caller's buffer is ordinary memory, and two
            // launches in flight may write the same frames from two XCDs -- written through (agent scope), so that no older line waits in
            // another L2 to be written back over this one (the reverb groups of the grid do the same)
            if (CH == 2) {
                const unsigned long long both = static_cast<unsigned long long>(__float_as_uint(o[0])) | (static_cast<unsigned long long>(__float_as_uint(o[CH - 1])) << 32);
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst + static_cast<size_t>(p) * 2), both, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                __hip_atomic_store(reinterpret_cast<unsigned*>(dst + p), __float_as_uint(o[0]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (last) {
            if (CH == 2) {
                *reinterpret_cast<float2*>(dst + static_cast<size_t>(p) * 2) = make_float2(o[0], o[CH - 1]);
            } else {
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    if (c < channels) dst[static_cast<size_t>(p) * channels + c] = o[c];
            }
        } else {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (c < channels) mixbuf[c * OALSFX_MAX_CHUNK + p] = o[c];
        }
    };
    // A body may lag by one tile (Fx::kLag == 1, the distortion): its call for tile k delivers the output of tile k - 1, and one
    // more call behind the last tile delivers the rest.  The mix so far of a tile is held back until its effect output arrives.
    constexpr int kLag = LagOf<Fx>::value;
    float held[CH];
    int held_pos = 0, held_L = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) held[c] = 0.0F;
    request(0);   // ahead of the effect's own start-up loads: one round trip for both
    fx.init(I);
    stamp();
    for (int base = 0; base < frames + (kLag ? 64 : 0); base += 64) {
        const int L = base < frames ? min(64, frames - base) : 0;
        const bool act = lane < L;
        const int pos = base + lane;
        float in[CH], win[CH], out[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) { in[c] = n_in[c]; win[c] = n_win[c]; out[c] = 0.0F; }
        float mix[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) mix[c] = n_mix[c];
        if (base + 64 < frames) request(base + 64);
        __builtin_amdgcn_sched_barrier(0); // keep the requests up here
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
            if (send_on) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float g = SRC.aux[slot].gains[c][k];
                    if (audible(g)) wet[k] += win[c] * g;
                }
            }
        }
        if (!first) {
#pragma unroll
            for (int c = 0; c < CH; ++c) out[c] = mix[c];
        }
        stamp();
        if constexpr (kLag != 0) {
            fx.template tile<CH>(I, wet, held, L, held_L);
            stamp();
            store_tile(held, held_pos, lane < held_L);
#pragma unroll
            for (int c = 0; c < CH; ++c) held[c] = out[c];
            held_pos = pos;
            held_L = L;
        } else {
            fx.template tile<CH>(I, wet, out, L);
            stamp();
            store_tile(out, pos, act);
        }
    }

    fx.finish(I);
    if (lane == 0) I.ss->seen_seq = I.sp->update_seq;
    if (first && !filtered && lane < channels) send_history_follow(ctx, inst, lane, channels, frames, src);
}


// The slots `slot` .. `slot + slot_count - 1` of one instance on one wavefront, in order (`lds`: kLdsFloats floats of this
// wavefront's own).  group.coop requires slot_count == 1 and the same effect type in all four wavefronts of the workgroup.
template <int CH>
__device__ __forceinline__ void wave_slots(const KernelCtx& ctx, int slot, int slot_count, int inst, int flags, float* lds, int lane, const Group& group)
{
    for (int sl = slot; sl < slot + slot_count; ++sl) {
        const int f = (flags & kFiltered) | ((flags & kFirst) && sl == slot ? kFirst : 0) | ((flags & kLast) && sl == slot + slot_count - 1 ? kLast : 0);
        const int type = __builtin_amdgcn_readfirstlane(ctx.params[static_cast<size_t>(inst) * ctx.slots + sl].type);
        KernelCtx c = ctx;
        c.wet_src = ctx.wet_src + static_cast<size_t>(sl - slot) * ctx.wet_plane;
        switch (type) {
        case OALSFX_NULL:
            if (f & (kFirst | kLast)) wave_instance<CH, NullW>(c, sl, inst, f, lds, lane, group); // a null effect in the middle does nothing
            break;
        case OALSFX_CHORUS:
        case OALSFX_FLANGER: wave_instance<CH, ModDelayW>(c, sl, inst, f, lds, lane, group); break;
        case OALSFX_COMPRESSOR: wave_instance<CH, CompressorW>(c, sl, inst, f, lds, lane, group); break;
        case OALSFX_DEDICATED_DIALOG:
        case OALSFX_DEDICATED_LFE: wave_instance<CH, DedicatedW>(c, sl, inst, f, lds, lane, group); break;
        case OALSFX_DISTORTION: wave_instance<CH, DistortionW>(c, sl, inst, f, lds, lane, group); break;
        case OALSFX_ECHO: wave_instance<CH, EchoW>(c, sl, inst, f, lds, lane, group); break;
        case OALSFX_EQUALIZER: wave_instance<CH, EqualizerW>(c, sl, inst, f, lds, lane, group); break;
        case OALSFX_RING_MODULATOR: wave_instance<CH, RingModW>(c, sl, inst, f, lds, lane, group); break;
        default: break;
        }
        wave_sync(); // the next slot of this instance reads the mix this one just wrote (same wavefront, program order)
    }
}

// The ring-light part of a grid: workgroup `block` of it, four wavefronts.  With segments (a single slot), the blocks follow
// the slot's type-sorted list segment by segment, so that a workgroup holds one effect type; without, wavefront w takes
// list[w].
// CHN: a launch of a run of chained launches (batch.cpp): the workgroup counts itself in, and a wavefront takes its
// instance when the launch before is through with it and hands it on behind its last store (common.hpp: turn_take, turn_hand_on).
template <int CH, bool CHN = false>
__device__ __forceinline__ void wave_block(const KernelCtx& ctx, int slot, int slot_count, const int* __restrict__ list, int count,
                                           const WaveSegments& seg, int flags, int block, float* lds_group, int lds_stride)
{
    const int lane = threadIdx.x & 63;
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (CHN && ctx.turn_started != nullptr && threadIdx.x == 0) __hip_atomic_fetch_add(ctx.turn_started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Group group{false, wib, block & 3, lds_group, lds_stride};
    int w = block * 4 + wib;
    if (seg.n > 0) {
        int k = 0, first_block = 0;
        for (; k + 1 < seg.n; ++k) {
            const int b = (seg.count[k] + 3) >> 2;
            if (block < first_block + b) break;
            first_block += b;
        }
        const int local = (block - first_block) * 4 + wib;
        group.coop = ((seg.coop_mask >> k) & 1u) != 0;
        if (local >= seg.count[k]) return; // never in a cooperative segment: those hold whole workgroups only
        w = seg.offset[k] + local;
    } else {
        // (a chained step: the list is the reverbs' grid's, and so is the order -- cu_major_position: an instance's wavefront of this launch
        // then comes up for a place on the chip when its workgroup of the reverbs' launch before gives one up, with its turn; in list order
        // the workgroups sat waiting for turns far down that grid's order, half the chip idle: 113 us per step instead of 94 in stream order)
        if (CHN && !((flags >> 8) & 8)) w = cu_major_position(block, (count + 3) >> 2) * 4 + wib; // (8: the experiment that takes the list as it comes)
        if (w >= count) return; // whole wavefronts leave; no workgroup barrier on this path
    }
    const int inst = __builtin_amdgcn_readfirstlane(list[w]);
    unsigned cu_before = 0;
    const size_t word = static_cast<size_t>(inst) * ctx.slots + ctx.turn_slot;
    if (CHN && ((flags >> 8) & 4) && ctx.turn != nullptr && ctx.turn_wait != 0u) {
        // Test switch 4 (tests/test_gpu_chained.py): the wavefront reads its instance's state and the first lines of its mix buffer *before*
        // its turn has come -- what the hand-over forbids -- so that this CU's L1 holds them as they were while the launch before is
        // still at work on them: without the acquire behind the wait (switch 2) the run must come out wrong, with it (1) right.
        unsigned junk = 0;
        for (int sl = slot; sl < slot + slot_count; ++sl) {
            const unsigned* st = reinterpret_cast<const unsigned*>(ctx.state + static_cast<size_t>(inst) * ctx.slots + sl);
#pragma unroll
            for (int k = 0; k < static_cast<int>(sizeof(SlotStateLines) / 256); ++k) {
                unsigned a;
                asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(st + 64 * k + lane) : "memory"); // (the wait inside: the compiler does not know the load is in flight)
                junk ^= a;
            }
        }
        if (ctx.mixbuf != nullptr) {
            const unsigned* mb = reinterpret_cast<const unsigned*>(ctx.mixbuf + static_cast<size_t>(inst) * ctx.channels * OALSFX_MAX_CHUNK);
            for (int c = 0; c < ctx.channels; ++c) {
                unsigned a;
                asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(a) : "v"(mb + c * OALSFX_MAX_CHUNK + 32 * lane) : "memory");
                junk ^= a;
            }
        }
        if (junk == 0x7E57AB1Eu && ctx.timeline) ctx.timeline[0] = junk; // (keeps the loads)
    }
    const bool mine = !CHN || turn_take(ctx, word, lane, flags >> 8, cu_before);
    // (its turn never came: the instance is left as it is -- and in a cooperative workgroup, whose wavefronts meet at barriers, all four are)
    if (CHN && group.coop) { if (__syncthreads_or(mine ? 0 : 1)) return; }
    else if (!mine) return;
    wave_slots<CH>(ctx, slot, slot_count, inst, flags, lds_group + wib * lds_stride, lane, group);
    if (CHN) turn_hand_on(ctx, word, lane, cu_before);
}

} // namespace wfx

} // namespace oalsfx_hip

#endif
