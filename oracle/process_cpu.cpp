// TEST INFRASTRUCTURE -- CPU oracle for the process path.  Not part of the product:
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
//
// A scalar, single-threaded restatement of what the reference does per buffer:
// Api::Impl::mix_data -> mix_source -> EffectState::process for every slot ->
// write_f32 (reference src/oalsfxpp.cpp:2917-3037, 3414-3431) and the ten
// do_process bodies (src/oalsfxpp.cpp:3952-7903).  It consumes the flat descriptors
// of include/oalsfx_desc.h (produced by the host update path) and keeps the effect
// state and delay rings in the same layout the HIP kernels use, so states can be
// compared word for word.
//
// Pinned against the compiled reference (oracle/_ref) bit-for-bit by
// tests/test_oracle_vs_reference.py and by the committed vectors in tests/golden/.
//
// Floating point: built with -O2 -ffp-contract=off; every expression keeps the
// reference's association.  sin() inside process loops goes through ref_sinf.h.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "oalsfx_desc.h"
#include "ref_sinf.h"

namespace {

constexpr int kMaxChunk = OALSFX_MAX_CHUNK;

struct Instance {
    int channels = 0;
    int slots = 0;
    oalsfx_source_params source{};
    oalsfx_source_state source_state{};
    oalsfx_slot_params params[OALSFX_MAX_SLOTS]{};
    oalsfx_slot_state state[OALSFX_MAX_SLOTS]{};
    std::vector<float> rings[OALSFX_MAX_SLOTS];
    // scratch (reference Device::sample_buffers_ / EffectSlot::wet_buffer_)
    float out[OALSFX_MAX_CHANNELS][kMaxChunk];
    float wet[OALSFX_MAX_SLOTS][OALSFX_EFFECT_CHANNELS][kMaxChunk];
};

inline bool audible(float gain) { return std::abs(gain) > OALSFX_SILENCE_GAIN; }

inline float lerp(float a, float b, float mu) { return a + ((b - a) * mu); }

// ---------------------------------------------------------------------------------------------
// Direct-form-I biquad over a stream (reference FilterState::process, src/oalsfxpp.cpp:984-1036).
// The reference special-cases the first two samples of a call only to fetch history; as a
// stream that is one recurrence with a two-sample shift register.
// ---------------------------------------------------------------------------------------------
inline float biquad_step(const oalsfx_biquad_t& c, oalsfx_hist_t& h, float x)
{
    const float y = (c.b0 * x) + (c.b1 * h.x[0]) + (c.b2 * h.x[1]) - (c.a1 * h.y[0]) - (c.a2 * h.y[1]);
    h.x[1] = h.x[0];
    h.x[0] = x;
    h.y[1] = h.y[0];
    h.y[0] = y;
    return y;
}

void biquad_run(const oalsfx_biquad_t& c, oalsfx_hist_t& h, int n, const float* src, float* dst)
{
    for (int i = 0; i < n; ++i) dst[i] = biquad_step(c, h, src[i]);
}

// process_pass_through (src/oalsfxpp.cpp:1038-1056): history follows the input on both sides.
void biquad_skip(oalsfx_hist_t& h, int n, const float* src)
{
    for (int i = 0; i < n; ++i) {
        h.x[1] = h.x[0];
        h.x[0] = src[i];
        h.y[1] = h.y[0];
        h.y[0] = src[i];
    }
}

// ---------------------------------------------------------------------------------------------
// Send front end (reference mix_source + apply_filters, src/oalsfxpp.cpp:2917-2982, 3101-3143)
// ---------------------------------------------------------------------------------------------
const float* send_filter(const oalsfx_send_params& p, oalsfx_hist_t& lp, oalsfx_hist_t& hp, int n, const float* src, float* tmp)
{
    switch (p.filter_type) {
    case OALSFX_AF_NONE:
        biquad_skip(lp, n, src);
        biquad_skip(hp, n, src);
        return src;
    case OALSFX_AF_LOW_PASS:
        biquad_run(p.lp, lp, n, src, tmp);
        biquad_skip(hp, n, tmp);
        return tmp;
    case OALSFX_AF_HIGH_PASS:
        biquad_skip(lp, n, src);
        biquad_run(p.hp, hp, n, src, tmp);
        return tmp;
    default:
        // both: low-pass then high-pass (the reference's 256-sample sub-chunks are invisible on a stream)
        for (int i = 0; i < n; ++i) tmp[i] = biquad_step(p.hp, hp, biquad_step(p.lp, lp, src[i]));
        return tmp;
    }
}

template <int N>
void add_scaled(float (*dst)[kMaxChunk], int out_channels, const float* gains, const float* data, int n)
{
    // MixHelpers::mix with counter 0: static gains, silent gains skipped (src/oalsfxpp.cpp:2752-2798)
    for (int c = 0; c < out_channels; ++c) {
        const float g = gains[c];
        if (!audible(g)) continue;
        for (int i = 0; i < n; ++i) dst[c][i] += data[i] * g;
    }
}

void mix_source(Instance& I, int n, const float* src)
{
    static thread_local float chan[kMaxChunk];
    static thread_local float tmp[kMaxChunk];
    const int ch = I.channels;
    for (int c = 0; c < ch; ++c) {
        for (int i = 0; i < n; ++i) chan[i] = src[(i * ch) + c];
        const float* s = send_filter(I.source.direct, I.source_state.lp[0][c], I.source_state.hp[0][c], n, chan, tmp);
        add_scaled<0>(I.out, I.source.direct.out_channels, I.source.direct.gains[c], s, n);
        for (int a = 0; a < I.slots; ++a) {
            const oalsfx_send_params& p = I.source.aux[a];
            if (p.out_channels == 0) continue; // null effect: send disabled (src/oalsfxpp.cpp:3355-3359)
            s = send_filter(p, I.source_state.lp[1 + a][c], I.source_state.hp[1 + a][c], n, chan, tmp);
            add_scaled<0>(I.wet[a], p.out_channels, p.gains[c], s, n);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Chorus / flanger (reference src/oalsfxpp.cpp:4113-4276)
// ---------------------------------------------------------------------------------------------
int lfo_delay(const oalsfx_moddelay_params& p, int phase)
{
    if (p.waveform == 1) return static_cast<int>((1.0F - std::abs(2.0F - (p.lfo_scale * phase))) * p.depth) + p.delay;
    return static_cast<int>(oracle_sinf(p.lfo_scale * phase) * p.depth) + p.delay;
}

void process_moddelay(const oalsfx_moddelay_params& p, oalsfx_moddelay_state& s, float* ring, int n,
                      const float (*wet)[kMaxChunk], float (*out)[kMaxChunk], int channels)
{
    float* side[2] = {ring, ring + p.ring_len};
    const int mask = p.ring_len - 1;
    static thread_local float tap[2][kMaxChunk];
    for (int i = 0; i < n; ++i) {
        // the LFO phase of each side restarts from offset % range every 128 samples in the reference,
        // which is the same as taking it modulo the range at every sample
        const int off = s.offset;
        const int phase[2] = {off % p.lfo_range, (off + p.lfo_disp) % p.lfo_range};
        for (int k = 0; k < 2; ++k) {
            float* buf = side[k];
            buf[off & mask] = wet[0][i];
            tap[k][i] = buf[(off - lfo_delay(p, phase[k])) & mask] * p.feedback;
            buf[off & mask] += tap[k][i];
        }
        s.offset = off + 1;
    }
    // the reference accumulates left then right per 128-sample chunk and per channel; per output sample
    // the order is the same: left tap, then right tap
    for (int c = 0; c < channels; ++c) {
        for (int base = 0; base < n; base += 128) {
            const int todo = std::min(128, n - base);
            for (int k = 0; k < 2; ++k) {
                const float g = p.gains[k][c];
                if (!audible(g)) continue;
                for (int i = base; i < base + todo; ++i) out[c][i] += tap[k][i] * g;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Compressor (reference src/oalsfxpp.cpp:4352-4453)
// ---------------------------------------------------------------------------------------------
void process_compressor(const oalsfx_compressor_params& p, oalsfx_compressor_state& s, int n,
                        const float (*wet)[kMaxChunk], float (*out)[kMaxChunk], int channels)
{
    static thread_local float t[4][kMaxChunk];
    float gc = s.gain_control;
    for (int i = 0; i < n; ++i) {
        float amplitude = 1.0F;
        if (p.enabled) {
            amplitude = std::abs(wet[0][i]);
            amplitude = std::max(amplitude + std::abs(wet[1][i]),
                                 std::max(amplitude + std::abs(wet[2][i]), amplitude + std::abs(wet[3][i])));
        }
        if (amplitude > gc) gc = std::min(gc + p.attack_rate, amplitude);
        else if (amplitude < gc) gc = std::max(gc - p.release_rate, amplitude);
        const float output = 1.0F / std::min(2.0F, std::max(0.5F, gc));
        for (int j = 0; j < 4; ++j) t[j][i] = wet[j][i] * output;
    }
    s.gain_control = gc;
    for (int j = 0; j < 4; ++j)
        for (int k = 0; k < channels; ++k) {
            const float g = p.gains[j][k];
            if (!audible(g)) continue;
            for (int i = 0; i < n; ++i) out[k][i] += g * t[j][i];
        }
}

// Per output sample the compressor adds B-format channel 0..3 in that order inside every 64-sample
// chunk; the loop above adds channel j over the whole buffer before j+1, which touches each out[k][i]
// in the same j order.  (The same holds for the other chunked effects below.)

// ---------------------------------------------------------------------------------------------
// Dedicated (reference src/oalsfxpp.cpp:4556-4576)
// ---------------------------------------------------------------------------------------------
void process_dedicated(const oalsfx_dedicated_params& p, int n, const float (*wet)[kMaxChunk], float (*out)[kMaxChunk], int channels)
{
    for (int c = 0; c < channels; ++c) {
        const float g = p.gains[c];
        if (!audible(g)) continue;
        for (int i = 0; i < n; ++i) out[c][i] += wet[0][i] * g;
    }
}

// ---------------------------------------------------------------------------------------------
// Distortion (reference src/oalsfxpp.cpp:4675-4750): 4x zero-stuffed oversampling
// ---------------------------------------------------------------------------------------------
void process_distortion(const oalsfx_distortion_params& p, oalsfx_distortion_state& s, int n,
                        const float (*wet)[kMaxChunk], float (*out)[kMaxChunk], int channels)
{
    static thread_local float dec[kMaxChunk];
    const float fc = p.edge_coeff;
    for (int i = 0; i < n; ++i) {
        for (int k = 0; k < 4; ++k) {
            const float in = (k == 0) ? wet[0][i] * 4.0F : 0.0F;
            float smp = biquad_step(p.low_pass, s.low_pass, in);
            smp = (1.0F + fc) * smp / (1.0F + (fc * std::abs(smp)));
            smp = (1.0F + fc) * smp / (1.0F + (fc * std::abs(smp))) * -1.0F;
            smp = (1.0F + fc) * smp / (1.0F + (fc * std::abs(smp)));
            const float y = biquad_step(p.band_pass, s.band_pass, smp);
            if (k == 0) dec[i] = y; // keep one sample out of four
        }
    }
    for (int c = 0; c < channels; ++c) {
        const float g = p.gains[c] * p.attenuation;
        if (!audible(g)) continue;
        for (int i = 0; i < n; ++i) out[c][i] += g * dec[i];
    }
}

// ---------------------------------------------------------------------------------------------
// Echo (reference src/oalsfxpp.cpp:4887-4962)
// ---------------------------------------------------------------------------------------------
void process_echo(const oalsfx_echo_params& p, oalsfx_echo_state& s, float* ring, int n,
                  const float (*wet)[kMaxChunk], float (*out)[kMaxChunk], int channels)
{
    static thread_local float tap[2][kMaxChunk];
    const int mask = p.ring_len - 1;
    for (int i = 0; i < n; ++i) {
        const int off = s.offset;
        tap[0][i] = ring[(off - p.tap1) & mask];
        tap[1][i] = ring[(off - p.tap2) & mask];
        // damping filter on second tap + input, then feedback into the line
        const float in = tap[1][i] + wet[0][i];
        const oalsfx_biquad_t& f = p.filter;
        oalsfx_hist_t& h = s.filter;
        const float y = (in * f.b0) + (h.x[0] * f.b1) + (h.x[1] * f.b2) - (h.y[0] * f.a1) - (h.y[1] * f.a2);
        h.x[1] = h.x[0]; h.x[0] = in;
        h.y[1] = h.y[0]; h.y[0] = y;
        ring[off & mask] = y * p.feed_gain;
        s.offset = off + 1;
    }
    for (int c = 0; c < channels; ++c)
        for (int base = 0; base < n; base += 128) {
            const int todo = std::min(128, n - base);
            for (int k = 0; k < 2; ++k) {
                const float g = p.gains[k][c];
                if (!audible(g)) continue;
                for (int i = base; i < base + todo; ++i) out[c][i] += tap[k][i] * g;
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Equalizer (reference src/oalsfxpp.cpp:5161-5213)
// ---------------------------------------------------------------------------------------------
void process_equalizer(const oalsfx_equalizer_params& p, oalsfx_equalizer_state& s, int n,
                       const float (*wet)[kMaxChunk], float (*out)[kMaxChunk], int channels)
{
    static thread_local float t[kMaxChunk];
    for (int ft = 0; ft < 4; ++ft) {
        for (int i = 0; i < n; ++i) {
            float v = wet[ft][i];
            for (int b = 0; b < 4; ++b) v = biquad_step(p.band[b], s.hist[b][ft], v);
            t[i] = v;
        }
        for (int k = 0; k < channels; ++k) {
            const float g = p.gains[ft][k];
            if (!audible(g)) continue;
            for (int i = 0; i < n; ++i) out[k][i] += g * t[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Ring modulator (reference src/oalsfxpp.cpp:5652-5784)
// ---------------------------------------------------------------------------------------------
float carrier(int waveform, int index)
{
    constexpr int frac_bits = 24;
    constexpr int frac_one = 1 << frac_bits;
    constexpr float tau = 6.28318530717958647692F;
    constexpr float pi = 3.14159265358979323846F;
    switch (waveform) {
    case 0: return oracle_sinf(index * (tau / frac_one) - pi) * 0.5F + 0.5F;
    case 1: return static_cast<float>(index) / frac_one;
    default: return static_cast<float>((index >> (frac_bits - 1)) & 1);
    }
}

void process_ringmod(const oalsfx_ringmod_params& p, oalsfx_ringmod_state& s, int n,
                     const float (*wet)[kMaxChunk], float (*out)[kMaxChunk], int channels)
{
    static thread_local float t[kMaxChunk];
    constexpr int frac_mask = (1 << 24) - 1;
    int end_index = s.index;
    for (int j = 0; j < 4; ++j) {
        int index = s.index; // every B-format channel sees the same carrier phase
        for (int i = 0; i < n; ++i) {
            const float f = biquad_step(p.filter, s.hist[j], wet[j][i]);
            index = (index + p.step) & frac_mask;
            t[i] = f * carrier(p.waveform, index);
        }
        end_index = index;
        for (int k = 0; k < channels; ++k) {
            const float g = p.gains[j][k];
            if (!audible(g)) continue;
            for (int i = 0; i < n; ++i) out[k][i] += g * t[i];
        }
    }
    s.index = end_index;
}

// ---------------------------------------------------------------------------------------------
// Reverb / EAX reverb (reference src/oalsfxpp.cpp:6078-6170 and 7358-7903)
// ---------------------------------------------------------------------------------------------
#ifdef OALSFX_ORACLE_PS
static long ps_tiles = 0, ps_used = 0; // experiment build: reverb blocks seen / evaluated the parallel-prefix way
#endif

struct Reverb {
    const oalsfx_reverb_params& p;
    oalsfx_reverb_state& s;
    float* slab;

    float* line(int ring, int j) const { return slab + p.ring_off[ring] + j * p.ring_len[ring]; }
    int mask(int ring) const { return p.ring_len[ring] - 1; }

    // delay_out_faded / delay_out_unfaded (src/oalsfxpp.cpp:7358-7399)
    float tap(bool faded, int ring, int j, int off0, int off1, float mu) const
    {
        const float* l = line(ring, j);
        const int m = mask(ring);
        if (!faded) return l[off0 & m];
        return lerp(l[off0 & m], l[off1 & m], mu);
    }

    // the scattering rotation (src/oalsfxpp.cpp:7510-7521)
    void scatter(float v[4]) const
    {
        const float x = p.mix_x, y = p.mix_y;
        const float f[4] = {v[0], v[1], v[2], v[3]};
        v[0] = (x * f[0]) + (y * (f[1] + -f[2] + f[3]));
        v[1] = (x * f[1]) + (y * (-f[0] + f[2] + f[3]));
        v[2] = (x * f[2]) + (y * (f[0] + -f[1] + f[3]));
        v[3] = (x * f[3]) + (y * (-f[0] + -f[1] + -f[2]));
    }

    // Gerzon vector all-pass (src/oalsfxpp.cpp:7533-7562)
    void allpass(bool faded, int ring, const int32_t* cur_off, const int32_t* new_off, float v[4], int offset, float mu) const
    {
        float f[4];
        for (int j = 0; j < 4; ++j) {
            const float input = v[j];
            v[j] = tap(faded, ring, j, offset - cur_off[j], offset - new_off[j], mu) - (p.ap_feed_coeff * input);
            f[j] = input + (p.ap_feed_coeff * v[j]);
        }
        scatter(f);
        for (int j = 0; j < 4; ++j) line(ring, j)[offset & mask(ring)] = f[j];
    }

    // folds a pending parameter update into the state: what do_update does to state variables
    // (modulator index rescale src/oalsfxpp.cpp:7028-7031, cross-fade trigger :6062-6075)
    void fold_update()
    {
        s.mod_index = static_cast<int>(s.mod_index * static_cast<int64_t>(p.mod_range) / s.mod_range);
        s.mod_range = p.mod_range;
        for (int j = 0; j < 4; ++j) {
            if (p.early_tap[j] != s.cur_early_tap[j] || p.early_ap_off[j] != s.cur_early_ap_off[j] ||
                p.early_line_off[j] != s.cur_early_line_off[j] || p.late_tap[j] != s.cur_late_tap[j] ||
                p.late_ap_off[j] != s.cur_late_ap_off[j] || p.late_line_off[j] != s.cur_late_line_off[j]) {
                s.fade_count = 0;
                break;
            }
        }
    }

    void block(int todo, bool faded, float fade0, const float (*a_format)[OALSFX_RV_MAX_UPDATE],
               float (*early_out)[OALSFX_RV_MAX_UPDATE], float (*late_out)[OALSFX_RV_MAX_UPDATE])
    {
        constexpr float fade_step = 1.0F / OALSFX_RV_FADE_SAMPLES;
        constexpr float tau = 6.28318530717958647692F;
        const int off0 = s.offset;

        // input shelves, then into the main delay (verb_pass / eax_verb_pass, src/oalsfxpp.cpp:7814-7903)
        for (int c = 0; c < 4; ++c) {
            float* main = line(OALSFX_RV_MAIN, c);
            for (int i = 0; i < todo; ++i) {
                float v = biquad_step(p.lp, s.lp[c], a_format[c][i]);
                if (p.is_eax) v = biquad_step(p.hp, s.hp[c], v);
                main[(off0 + i) & mask(OALSFX_RV_MAIN)] = v;
            }
        }

        // early reflections (src/oalsfxpp.cpp:7625-7672)
        {
            float fade = fade0;
            for (int i = 0; i < todo; ++i) {
                const int off = off0 + i;
                float f[4];
                for (int j = 0; j < 4; ++j)
                    f[j] = tap(faded, OALSFX_RV_MAIN, j, off - s.cur_early_tap[j], off - p.early_tap[j], fade) * p.early_tap_coeff[j];
                allpass(faded, OALSFX_RV_EARLY_AP, s.cur_early_ap_off, p.early_ap_off, f, off, fade);
                for (int j = 0; j < 4; ++j) line(OALSFX_RV_EARLY_LINE, j)[off & mask(OALSFX_RV_EARLY_LINE)] = f[3 - j];
                for (int j = 0; j < 4; ++j)
                    f[j] += tap(faded, OALSFX_RV_EARLY_LINE, j, off - s.cur_early_line_off[j], off - p.early_line_off[j], fade) * p.early_line_coeff[j];
                for (int j = 0; j < 4; ++j) early_out[j][i] = f[j];
                std::swap(f[0], f[3]);
                std::swap(f[1], f[2]);
                scatter(f);
                for (int j = 0; j < 4; ++j) line(OALSFX_RV_MAIN, j)[(off - p.late_feed_tap) & mask(OALSFX_RV_MAIN)] = f[j];
                fade += fade_step;
            }
        }

        // late reverb (src/oalsfxpp.cpp:7735-7794), modulation delays first (src/oalsfxpp.cpp:7443-7470)
        {
            int moddelay[OALSFX_RV_MAX_UPDATE];
            int index = s.mod_index;
            float range = s.mod_filter;
            for (int i = 0; i < todo; ++i) {
                const float sinus = oracle_sinf(tau * index / s.mod_range);
                index = (index + 1) % s.mod_range;
                range = lerp(range, p.mod_depth, p.mod_coeff);
                moddelay[i] = static_cast<int>(std::lround(range * sinus));
            }
            s.mod_index = index;
            s.mod_filter = range;

#ifdef OALSFX_ORACLE_PS
            // EXPERIMENT (scripts/ps_error.py), never the parity build: the two first-order T60 sections evaluated the way a
            // wavefront parallel prefix would, 64 samples at a time -- each sample an affine map x -> a x + b, composed in six
            // doubling steps in fp32 (BASELINE's north_star suggests this for the first-order sections); everything else as the
            // reference.  Only where a tile's inputs lie entirely before it (steady state, taps of 64 samples or more).
            float ps_out[4][OALSFX_RV_MAX_UPDATE];
            bool ps = !faded;
            for (int j = 0; j < 4; ++j)
                ps = ps && s.cur_late_tap[j] - p.late_feed_tap >= 64 && s.cur_late_tap[j] >= 64 && s.cur_late_line_off[j] - 64 >= 64;
            ps_tiles += 1;
            if (ps) {
                ps_used += 1;
                for (int base = 0; base < todo; base += 64) {
                    const int n = std::min(64, todo - base);
                    for (int j = 0; j < 4; ++j) {
                        float u[64], a[64], b[64], na[64], nb[64];
                        for (int i = 0; i < n; ++i) {
                            const int off = off0 + base + i;
                            u[i] = line(OALSFX_RV_MAIN, j)[(off - s.cur_late_tap[j]) & mask(OALSFX_RV_MAIN)] * p.density_gain;
                            u[i] += line(OALSFX_RV_LATE_LINE, j)[(off - moddelay[base + i] - s.cur_late_line_off[j]) & mask(OALSFX_RV_LATE_LINE)];
                        }
                        float* st = &s.t60[j][0][0];
                        for (int section = 0; section < 2; ++section) {
                            const float* c = section ? p.t60_hf[j] : p.t60_lf[j];
                            const float x_before = st[2 * section], y_before = st[2 * section + 1];
                            for (int i = 0; i < n; ++i) {
                                a[i] = c[2];
                                b[i] = (c[0] * u[i]) + (c[1] * (i ? u[i - 1] : x_before)); // the feed-forward half, per lane
                            }
                            for (int d = 1; d < 64; d <<= 1) {
                                for (int i = 0; i < n; ++i) {
                                    na[i] = a[i]; nb[i] = b[i];
                                    if (i >= d) { na[i] = a[i] * a[i - d]; nb[i] = (a[i] * b[i - d]) + b[i]; }
                                }
                                for (int i = 0; i < n; ++i) { a[i] = na[i]; b[i] = nb[i]; }
                            }
                            st[2 * section] = u[n - 1];
                            for (int i = 0; i < n; ++i) u[i] = (a[i] * y_before) + b[i]; // the section's outputs feed the next one
                            st[2 * section + 1] = u[n - 1];
                        }
                        for (int i = 0; i < n; ++i) ps_out[j][base + i] = p.t60_mid[j] * u[i];
                    }
                }
            }
#endif
            float fade = fade0;
            for (int i = 0; i < todo; ++i) {
                const int off = off0 + i;
                float f[4];
                for (int j = 0; j < 4; ++j)
                    f[j] = tap(faded, OALSFX_RV_MAIN, j, off - s.cur_late_tap[j], off - p.late_tap[j], fade) * p.density_gain;
                const int delayed = off - moddelay[i];
                for (int j = 0; j < 4; ++j)
                    f[j] += tap(faded, OALSFX_RV_LATE_LINE, j, delayed - s.cur_late_line_off[j], delayed - p.late_line_off[j], fade);
#ifdef OALSFX_ORACLE_PS
                if (ps) {
                    for (int j = 0; j < 4; ++j) f[j] = ps_out[j][i];
                } else
#endif
                for (int j = 0; j < 4; ++j) {
                    // two first-order sections then the mid-band gain (src/oalsfxpp.cpp:7691-7719)
                    float* st = &s.t60[j][0][0];
                    const float o1 = (p.t60_lf[j][0] * f[j]) + (p.t60_lf[j][1] * st[0]) + (p.t60_lf[j][2] * st[1]);
                    st[0] = f[j];
                    st[1] = o1;
                    const float o2 = (p.t60_hf[j][0] * o1) + (p.t60_hf[j][1] * st[2]) + (p.t60_hf[j][2] * st[3]);
                    st[2] = o1;
                    st[3] = o2;
                    f[j] = p.t60_mid[j] * o2;
                }
                allpass(faded, OALSFX_RV_LATE_AP, s.cur_late_ap_off, p.late_ap_off, f, off, fade);
                for (int j = 0; j < 4; ++j) late_out[j][i] = f[j];
                std::swap(f[0], f[3]);
                std::swap(f[1], f[2]);
                scatter(f);
                for (int j = 0; j < 4; ++j) line(OALSFX_RV_LATE_LINE, j)[off & mask(OALSFX_RV_LATE_LINE)] = f[j];
                fade += fade_step;
            }
        }
        s.offset = off0 + todo;
    }
};

// MixHelpers::mix with a gain ramp over `counter` samples (src/oalsfxpp.cpp:2752-2798)
void ramp_mix(const float* data, int channels, float (*out)[kMaxChunk], float* current, const float* target, int counter, int pos0, int size)
{
    const float delta = (counter > 0) ? 1.0F / static_cast<float>(counter) : 0.0F;
    for (int c = 0; c < channels; ++c) {
        int pos = 0;
        float gain = current[c];
        const float step = (target[c] - gain) * delta;
        if (std::abs(step) > std::numeric_limits<float>::epsilon()) {
            const int ramp = std::min(size, counter);
            for (; pos < ramp; ++pos) {
                out[c][pos0 + pos] += data[pos] * gain;
                gain += step;
            }
            if (pos == counter) gain = target[c];
            current[c] = gain;
        }
        if (!audible(gain)) continue;
        for (; pos < size; ++pos) out[c][pos0 + pos] += data[pos] * gain;
    }
}

void process_reverb(const oalsfx_slot_params& sp, oalsfx_slot_state& ss, float* slab, int n,
                    const float (*wet)[kMaxChunk], float (*out)[kMaxChunk], int channels)
{
    // B-format -> A-format (reference b2a, src/oalsfxpp.cpp:6377-6383)
    static const float b2a[4][4] = {
        {0.288675134595F, 0.288675134595F, 0.288675134595F, 0.288675134595F},
        {0.288675134595F, -0.288675134595F, -0.288675134595F, 0.288675134595F},
        {0.288675134595F, 0.288675134595F, -0.288675134595F, -0.288675134595F},
        {0.288675134595F, -0.288675134595F, 0.288675134595F, -0.288675134595F},
    };
    Reverb rv{sp.u.reverb, ss.u.reverb, slab};
    if (ss.seen_seq != sp.update_seq) {
        rv.fold_update();
        ss.seen_seq = sp.update_seq;
    }
    oalsfx_reverb_state& s = ss.u.reverb;
    const oalsfx_reverb_params& p = sp.u.reverb;
    float a_format[4][OALSFX_RV_MAX_UPDATE], early[4][OALSFX_RV_MAX_UPDATE], late[4][OALSFX_RV_MAX_UPDATE];

    for (int base = 0; base < n;) {
        int todo = std::min(n - base, OALSFX_RV_MAX_UPDATE);
        if (OALSFX_RV_FADE_SAMPLES - s.fade_count > 0) todo = std::min(todo, OALSFX_RV_FADE_SAMPLES - s.fade_count);
        const float fade = static_cast<float>(s.fade_count) / OALSFX_RV_FADE_SAMPLES;

        for (int c = 0; c < 4; ++c) {
            for (int i = 0; i < todo; ++i) a_format[c][i] = 0.0F;
            for (int k = 0; k < 4; ++k) {
                if (!audible(b2a[c][k])) continue;
                for (int i = 0; i < todo; ++i) a_format[c][i] += wet[k][base + i] * b2a[c][k];
            }
        }

        rv.block(todo, fade < 1.0F, fade, a_format, early, late);

        if (s.fade_count < OALSFX_RV_FADE_SAMPLES) {
            s.fade_count += todo;
            if (s.fade_count >= OALSFX_RV_FADE_SAMPLES) {
                s.fade_count = OALSFX_RV_FADE_SAMPLES;
                for (int c = 0; c < 4; ++c) {
                    s.cur_early_tap[c] = p.early_tap[c];
                    s.cur_early_ap_off[c] = p.early_ap_off[c];
                    s.cur_early_line_off[c] = p.early_line_off[c];
                    s.cur_late_tap[c] = p.late_tap[c];
                    s.cur_late_ap_off[c] = p.late_ap_off[c];
                    s.cur_late_line_off[c] = p.late_line_off[c];
                }
            }
        }

        for (int c = 0; c < 4; ++c) ramp_mix(early[c], channels, out, s.early_cur_gain[c], p.early_pan[c], n - base, base, todo);
        for (int c = 0; c < 4; ++c) ramp_mix(late[c], channels, out, s.late_cur_gain[c], p.late_pan[c], n - base, base, todo);
        base += todo;
    }
}

// ---------------------------------------------------------------------------------------------
// One chunk of at most 2048 frames (reference mix_data, src/oalsfxpp.cpp:2984-3037)
// ---------------------------------------------------------------------------------------------
void mix_chunk(Instance& I, int n, const float* src, float* dst)
{
    for (int c = 0; c < I.channels; ++c) std::fill_n(I.out[c], n, 0.0F);
    for (int a = 0; a < I.slots; ++a)
        for (int c = 0; c < OALSFX_EFFECT_CHANNELS; ++c) std::fill_n(I.wet[a][c], n, 0.0F);

    mix_source(I, n, src);

    for (int a = 0; a < I.slots; ++a) {
        const oalsfx_slot_params& p = I.params[a];
        oalsfx_slot_state& s = I.state[a];
        float* ring = I.rings[a].empty() ? nullptr : I.rings[a].data();
        switch (p.type) {
        case OALSFX_CHORUS:
        case OALSFX_FLANGER: process_moddelay(p.u.moddelay, s.u.moddelay, ring, n, I.wet[a], I.out, I.channels); break;
        case OALSFX_COMPRESSOR: process_compressor(p.u.compressor, s.u.compressor, n, I.wet[a], I.out, I.channels); break;
        case OALSFX_DEDICATED_DIALOG:
        case OALSFX_DEDICATED_LFE: process_dedicated(p.u.dedicated, n, I.wet[a], I.out, I.channels); break;
        case OALSFX_DISTORTION: process_distortion(p.u.distortion, s.u.distortion, n, I.wet[a], I.out, I.channels); break;
        case OALSFX_ECHO: process_echo(p.u.echo, s.u.echo, ring, n, I.wet[a], I.out, I.channels); break;
        case OALSFX_EQUALIZER: process_equalizer(p.u.equalizer, s.u.equalizer, n, I.wet[a], I.out, I.channels); break;
        case OALSFX_RING_MODULATOR: process_ringmod(p.u.ringmod, s.u.ringmod, n, I.wet[a], I.out, I.channels); break;
        case OALSFX_REVERB:
        case OALSFX_EAX_REVERB: process_reverb(p, s, ring, n, I.wet[a], I.out, I.channels); break;
        default: break; // null effect
        }
        s.seen_seq = p.update_seq;
    }

    for (int c = 0; c < I.channels; ++c)
        for (int i = 0; i < n; ++i) dst[(i * I.channels) + c] = I.out[c][i];
}

int ring_floats(const oalsfx_slot_params& p)
{
    switch (p.type) {
    case OALSFX_CHORUS:
    case OALSFX_FLANGER: return 2 * p.u.moddelay.ring_len;
    case OALSFX_ECHO: return p.u.echo.ring_len;
    case OALSFX_REVERB:
    case OALSFX_EAX_REVERB: {
        int n = 0;
        for (int r = 0; r < 5; ++r) n += 4 * p.u.reverb.ring_len[r];
        return n;
    }
    default: return 0;
    }
}

} // namespace

extern "C" {

void* oracle_create(int channels, int slots)
{
    auto* I = new Instance{};
    I->channels = channels;
    I->slots = slots;
    return I;
}

void oracle_destroy(void* h) { delete static_cast<Instance*>(h); }

void oracle_set_source(void* h, const oalsfx_source_params* p) { static_cast<Instance*>(h)->source = *p; }

// `restart` != 0: the effect type changed -> fresh state and zeroed rings (reference
// EffectSlot::set_effect + do_construct/do_update_device, src/oalsfxpp.cpp:2688-2709)
void oracle_set_slot(void* h, int slot, const oalsfx_slot_params* p, int restart)
{
    Instance& I = *static_cast<Instance*>(h);
    I.params[slot] = *p;
    if (restart) {
        std::memset(&I.state[slot], 0, sizeof(oalsfx_slot_state));
        if (p->type == OALSFX_COMPRESSOR) I.state[slot].u.compressor.gain_control = 1.0F;
        if (p->type == OALSFX_REVERB || p->type == OALSFX_EAX_REVERB) I.state[slot].u.reverb.mod_range = 1;
        I.state[slot].seen_seq = p->update_seq - 1;
        I.rings[slot].assign(static_cast<size_t>(ring_floats(*p)), 0.0F);
    }
}

void oracle_mix(void* h, int frames, const float* src, float* dst)
{
    Instance& I = *static_cast<Instance*>(h);
    for (int done = 0; done < frames;) {
        const int n = std::min(frames - done, kMaxChunk); // Api::mix chunking, src/oalsfxpp.cpp:3818-3826
        mix_chunk(I, n, src + static_cast<size_t>(done) * I.channels, dst + static_cast<size_t>(done) * I.channels);
        done += n;
    }
}

void oracle_get_state(void* h, int slot, oalsfx_slot_state* out) { *out = static_cast<Instance*>(h)->state[slot]; }
void oracle_get_source_state(void* h, oalsfx_source_state* out) { *out = static_cast<Instance*>(h)->source_state; }

int oracle_get_ring(void* h, int slot, float* out, int max_floats)
{
    const auto& r = static_cast<Instance*>(h)->rings[slot];
    const int n = std::min(static_cast<int>(r.size()), max_floats);
    if (out) std::copy(r.begin(), r.begin() + n, out);
    return static_cast<int>(r.size());
}

// Synthetic input shared with bench.py / the GPU generator: xorshift32 seeded per
// (instance, buffer), uniform in [-1, 1).
void oracle_synth(uint32_t instance, uint32_t buffer_index, int count, float* out)
{
    uint32_t x = 0x9E3779B9u ^ (instance * 2654435761u) ^ buffer_index;
    if (x == 0) x = 1;
    for (int i = 0; i < count; ++i) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        out[i] = static_cast<float>(x >> 8) * (1.0F / 8388608.0F) - 1.0F;
    }
}

// CPU baseline: `n_instances` copies of the prototype instance `h` (same parameters, own state) each
// advanced by `buffers` buffers of `frames` synthetic frames on `threads` threads.  Returns seconds.
double oracle_bench(void* h, int n_instances, int frames, int warmup, int buffers, int threads)
{
    const Instance& proto = *static_cast<Instance*>(h);
    std::vector<Instance*> inst(n_instances);
    for (auto& p : inst) p = new Instance(proto);
    auto run = [&](int first_buf, int count) {
        std::atomic<int> next{0};
        auto worker = [&]() {
            std::vector<float> src(static_cast<size_t>(frames) * proto.channels), dst(src.size());
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= n_instances) break;
                for (int b = 0; b < count; ++b) {
                    oracle_synth(static_cast<uint32_t>(i), static_cast<uint32_t>(first_buf + b), static_cast<int>(src.size()), src.data());
                    oracle_mix(inst[i], frames, src.data(), dst.data());
                }
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; ++t) pool.emplace_back(worker);
        worker();
        for (auto& t : pool) t.join();
    };
    run(0, warmup);
    const auto t0 = std::chrono::steady_clock::now();
    run(warmup, buffers);
    const auto t1 = std::chrono::steady_clock::now();
    for (auto p : inst) delete p;
    return std::chrono::duration<double>(t1 - t0).count();
}

#ifdef OALSFX_ORACLE_PS
void oracle_ps_counts(long* blocks, long* used) { *blocks = ps_tiles; *used = ps_used; }
#endif

} // extern "C"
