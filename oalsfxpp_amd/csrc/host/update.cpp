// Parameter-update path: effect properties -> flat kernel descriptors.
//
// One function per effect family; each follows the arithmetic of the reference's
// do_update_device / do_update for that effect so that the derived floats and
// integer taps are identical (checked byte-for-byte against dumps of the compiled
// reference in tests/test_update_path.py).
#include <cmath>
#include <cstring>

#include "core.hpp"

namespace oalsfx_host {

using oalsfxpp::Effect;
using oalsfxpp::EffectProps;
using oalsfxpp::EffectType;
using oalsfxpp::SendProps;

namespace {

const float identity4[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};

// B-format channel i -> speaker gains through the first-order decoder (used by the
// compressor, equalizer and ring modulator; reference src/oalsfxpp.cpp:4341-4349).
void bformat_gains(const DeviceDesc& dev, float out[4][OALSFX_MAX_CHANNELS])
{
    for (int i = 0; i < 4; ++i) first_order_gains_foa(dev, identity4[i], 1.0F, out[i]);
}

// ---- chorus / flanger (reference src/oalsfxpp.cpp:4021-4111, 5292-5382) ----
template <typename P>
void derive_moddelay(const DeviceDesc& dev, const P& props, float max_delay, oalsfx_moddelay_params& o)
{
    const float frequency = static_cast<float>(dev.rate);
    o.ring_len = next_power_of_2(static_cast<int>(max_delay * 2.0F * dev.rate) + 1);
    o.waveform = props.waveform_;
    o.feedback = props.feedback_;
    o.delay = static_cast<int>(props.delay_ * frequency);
    o.depth = props.depth_ * o.delay; // LFO depth relative to the sample delay

    float coeffs[max_ambi_coeffs];
    calc_angle_coeffs(-pi_2, 0.0F, 0.0F, coeffs);
    panning_gains_dry(dev, coeffs, 1.0F, o.gains[0]);
    calc_angle_coeffs(pi_2, 0.0F, 0.0F, coeffs);
    panning_gains_dry(dev, coeffs, 1.0F, o.gains[1]);

    const int phase = props.phase_;
    const float rate = props.rate_;
    if (!(rate > 0.0F)) {
        o.lfo_scale = 0.0F;
        o.lfo_range = 1;
        o.lfo_disp = 0;
    } else {
        o.lfo_range = static_cast<int>(frequency / rate + 0.5F);
        o.lfo_scale = (o.waveform == P::waveform_triangle ? 4.0F : tau) / o.lfo_range;
        if (phase >= 0) o.lfo_disp = static_cast<int>(o.lfo_range * (phase / 360.0F));
        else o.lfo_disp = static_cast<int>(o.lfo_range * ((360 + phase) / 360.0F));
    }
}

// ---- compressor (reference src/oalsfxpp.cpp:4319-4350) ----
void derive_compressor(const DeviceDesc& dev, const EffectProps::Compressor& props, oalsfx_compressor_params& o)
{
    const float attack_time = dev.rate * 0.2F;
    const float release_time = dev.rate * 0.4F;
    o.attack_rate = 1.0F / attack_time;
    o.release_rate = 1.0F / release_time;
    o.enabled = props.on_off_ ? 1 : 0;
    bformat_gains(dev, o.gains);
}

// ---- dedicated dialog / LFE (reference src/oalsfxpp.cpp:4509-4554) ----
// Device::get_channel_index searches an empty range (src/oalsfxpp.cpp:2577-2578) and always answers
// "not found": the LFE variant therefore stays silent and dialog is always panned to the front.
void derive_dedicated(const DeviceDesc& dev, EffectType type, const EffectProps::Dedicated& props, oalsfx_dedicated_params& o)
{
    for (float& g : o.gains) g = 0.0F;
    if (type == EffectType::dedicated_dialog) {
        float coeffs[max_ambi_coeffs];
        calc_angle_coeffs(0.0F, 0.0F, 0.0F, coeffs);
        panning_gains_dry(dev, coeffs, props.gain_, o.gains);
    }
}

// ---- distortion (reference src/oalsfxpp.cpp:4627-4673) ----
void derive_distortion(const DeviceDesc& dev, const EffectProps::Distortion& props, oalsfx_distortion_params& o)
{
    const float frequency = static_cast<float>(dev.rate);
    o.attenuation = props.gain_;

    float edge = std::sin(props.edge_ * pi_2);
    edge = std::min(edge, 0.99F);
    o.edge_coeff = 2.0F * edge / (1.0F - edge);

    // the filters run on the 4x oversampled signal
    float cutoff = props.low_pass_cutoff_;
    float bandwidth = (cutoff / 2.0F) / (cutoff * 0.67F);
    design_biquad(FilterKind::low_pass, 1.0F, cutoff / (frequency * 4.0F),
                  rcp_q_from_bandwidth(cutoff / (frequency * 4.0F), bandwidth), o.low_pass);

    cutoff = props.eq_center_;
    bandwidth = props.eq_bandwidth_ / (cutoff * 0.67F);
    design_biquad(FilterKind::band_pass, 1.0F, cutoff / (frequency * 4.0F),
                  rcp_q_from_bandwidth(cutoff / (frequency * 4.0F), bandwidth), o.band_pass);

    ambient_gains_dry(dev, 1.0F, o.gains);
}

// ---- echo (reference src/oalsfxpp.cpp:4817-4885) ----
void derive_echo(const DeviceDesc& dev, const EffectProps::Echo& props, oalsfx_echo_params& o)
{
    const int frequency = dev.rate;
    int maxlen = static_cast<int>(EffectProps::Echo::max_delay * frequency) + 1;
    maxlen += static_cast<int>(EffectProps::Echo::max_lr_delay * frequency) + 1;
    o.ring_len = next_power_of_2(maxlen);

    o.tap1 = static_cast<int>(props.delay_ * frequency) + 1;
    o.tap2 = static_cast<int>(props.lr_delay_ * frequency);
    o.tap2 += o.tap1;

    float spread = props.spread_;
    const float lrpan = (spread < 0.0F) ? -1.0F : 1.0F;
    // echo spread (0 omni, +-1 directional) -> coverage angle (0 point, tau omni)
    spread = std::asin(1.0F - std::abs(spread)) * 4.0F;

    o.feed_gain = props.feedback_;

    const float damp_gain = std::max(1.0F - props.damping_, 0.0625F);
    design_biquad(FilterKind::high_shelf, damp_gain, SendProps::lp_frequency_reference / frequency,
                  rcp_q_from_slope(damp_gain, 1.0F), o.filter);

    float coeffs[max_ambi_coeffs];
    calc_angle_coeffs(-pi_2 * lrpan, 0.0F, spread, coeffs);
    panning_gains_dry(dev, coeffs, 1.0F, o.gains[0]);
    calc_angle_coeffs(pi_2 * lrpan, 0.0F, spread, coeffs);
    panning_gains_dry(dev, coeffs, 1.0F, o.gains[1]);
}

// ---- equalizer (reference src/oalsfxpp.cpp:5076-5159) ----
void derive_equalizer(const DeviceDesc& dev, const EffectProps::Equalizer& props, oalsfx_equalizer_params& o)
{
    const float frequency = static_cast<float>(dev.rate);
    bformat_gains(dev, o.gains);

    float gain = std::max(std::sqrt(props.low_gain_), 0.0625F);
    float freq_mult = props.low_cutoff_ / frequency;
    design_biquad(FilterKind::low_shelf, gain, freq_mult, rcp_q_from_slope(gain, 0.75F), o.band[0]);

    gain = std::max(props.mid1_gain_, 0.0625F);
    freq_mult = props.mid1_center_ / frequency;
    design_biquad(FilterKind::peaking, gain, freq_mult, rcp_q_from_bandwidth(freq_mult, props.mid1_width_), o.band[1]);

    gain = std::max(props.mid2_gain_, 0.0625F);
    freq_mult = props.mid2_center_ / frequency;
    design_biquad(FilterKind::peaking, gain, freq_mult, rcp_q_from_bandwidth(freq_mult, props.mid2_width_), o.band[2]);

    gain = std::max(std::sqrt(props.high_gain_), 0.0625F);
    freq_mult = props.high_cutoff_ / frequency;
    design_biquad(FilterKind::high_shelf, gain, freq_mult, rcp_q_from_slope(gain, 0.75F), o.band[3]);
}

// ---- ring modulator (reference src/oalsfxpp.cpp:5598-5650) ----
void derive_ringmod(const DeviceDesc& dev, const EffectProps::RingModulator& props, oalsfx_ringmod_params& o)
{
    constexpr int frac_one = 1 << 24;
    o.waveform = (props.waveform_ == EffectProps::RingModulator::waveform_sinusoid) ? 0
               : (props.waveform_ == EffectProps::RingModulator::waveform_sawtooth) ? 1 : 2;
    o.step = static_cast<int>(props.frequency_ * frac_one / dev.rate);
    if (o.step == 0) o.step = 1;

    // one-pole high-pass expressed through the biquad slots
    const float cw = std::cos(tau * props.high_pass_cutoff_ / dev.rate);
    const float a = (2.0F - cw) - std::sqrt(std::pow(2.0F - cw, 2.0F) - 1.0F);
    o.filter.b0 = a;
    o.filter.b1 = -a;
    o.filter.b2 = 0.0F;
    o.filter.a1 = -a;
    o.filter.a2 = 0.0F;
    bformat_gains(dev, o.gains);
}

// ---------------------------------------------------------------------------------------------
// Reverb / EAX reverb (reference src/oalsfxpp.cpp:5928-6076 and the helpers at :6538-7350)
// ---------------------------------------------------------------------------------------------
namespace rv {

constexpr float speed_of_sound_mps = 343.3F;
constexpr float decay_gain = 0.001F; // -60 dB target of the decay time
constexpr float line_multiplier = 9.0F;
constexpr float early_tap_lengths[4] = {0.000000E+0F, 1.010676E-3F, 2.126553E-3F, 3.358580E-3F};
constexpr float early_allpass_lengths[4] = {4.854840E-4F, 5.360178E-4F, 5.918117E-4F, 6.534130E-4F};
constexpr float early_line_lengths[4] = {2.992520E-3F, 5.456575E-3F, 7.688329E-3F, 9.709681E-3F};
constexpr float late_allpass_lengths[4] = {8.091400E-4F, 1.019453E-3F, 1.407968E-3F, 1.618280E-3F};
constexpr float late_line_lengths[4] = {9.709681E-3F, 1.223343E-2F, 1.689561E-2F, 1.941936E-2F};
constexpr float modulation_depth_coeff = 1.0F / 4096.0F;
constexpr float modulation_filter_coeff = 0.048F;
constexpr float modulation_filter_const = 100000.0F;

using R = EffectProps::Reverb;

int line_samples(float length, int frequency, int extra)
{
    // power-of-two ring covering `length` seconds rounded up, plus `extra` frames
    const int n = static_cast<int>(std::ceil(length * frequency));
    return next_power_of_2(n + extra);
}

void ring_lengths(int frequency, int len[5])
{
    const float multiplier = 1.0F + line_multiplier;
    float length = R::max_reflections_delay + (early_tap_lengths[3] * multiplier) + R::max_late_reverb_delay +
                   ((late_line_lengths[3] - late_line_lengths[0]) * 0.25F * multiplier);
    len[OALSFX_RV_MAIN] = line_samples(length, frequency, OALSFX_RV_MAX_UPDATE);
    len[OALSFX_RV_EARLY_AP] = line_samples(early_allpass_lengths[3] * multiplier, frequency, 0);
    len[OALSFX_RV_EARLY_LINE] = line_samples(early_line_lengths[3] * multiplier, frequency, 0);
    len[OALSFX_RV_LATE_AP] = line_samples(late_allpass_lengths[3] * multiplier, frequency, 0);
    length = std::max(R::max_echo_time, late_line_lengths[3] * multiplier) +
             (R::max_modulation_time * modulation_depth_coeff / 2.0F);
    len[OALSFX_RV_LATE_LINE] = line_samples(length, frequency, 0);
}

float decay_coeff(float length, float decay_time) { return std::pow(decay_gain, length / decay_time); }

float decay_length(float coeff, float decay_time) { return std::log10(coeff) * decay_time / std::log10(decay_gain); }

float density_gain(float a) { return std::sqrt(1.0F - (a * a)); }

float limited_hf_ratio(float hf_ratio, float air_absorption_gain_hf, float decay_time)
{
    const float limit_ratio = 1.0F / (decay_length(air_absorption_gain_hf, decay_time) * speed_of_sound_mps);
    return clamp(limit_ratio, 0.1F, hf_ratio);
}

void pass_through(float c[3]) { c[0] = 1.0F; c[1] = 0.0F; c[2] = 0.0F; }

// first-order sections of the T60 filter, c = {c0 (x[n]), c1 (x[n-1]), c2 (y[n-1])}
void highpass_coeffs(float gain, float w, float c[3])
{
    if (gain >= 1.0F) return pass_through(c);
    const float g = std::max(0.001F, gain);
    const float g2 = g * g;
    const float cw = std::cos(w);
    const float p = g / ((g * cw) + std::sqrt((cw - 1.0F) * ((g2 * cw) + g2 - 2.0F)));
    c[0] = p; c[1] = -p; c[2] = p;
}

void lowpass_coeffs(float gain, float w, float c[3])
{
    if (gain >= 1.0F) return pass_through(c);
    const float g = std::max(0.001F, gain);
    const float g2 = g * g;
    const float cw = std::cos(w);
    const float a = (1.0F - (g2 * cw) - std::sqrt((2.0F * g2 * (1.0F - cw)) - (g2 * g2 * (1.0F - (cw * cw))))) / (1.0F - g2);
    c[0] = 1.0F - a; c[1] = 0.0F; c[2] = a;
}

void shelf_common(float g, float& alpha, float& beta0, float& beta1)
{
    const float n = (g + 1.0F) / (g - 1.0F);
    alpha = n + std::sqrt((n * n) - 1.0F);
    beta0 = (1.0F + g + (1.0F - g) * alpha) / 2.0F;
    beta1 = (1.0F - g + (1.0F + g) * alpha) / 2.0F;
}

void low_shelf_coeffs(float gain, float w, float c[3])
{
    if (gain >= 1.0F) return pass_through(c);
    const float g = std::max(0.001F, gain);
    const float rw = pi - w;
    const float p = std::sin((0.5F * rw) - (0.25F * pi)) / std::sin((0.5F * rw) + (0.25F * pi));
    float alpha, beta0, beta1;
    shelf_common(g, alpha, beta0, beta1);
    c[0] = (beta0 + (p * beta1)) / (1.0F + (p * alpha));
    c[1] = -(beta1 + (p * beta0)) / (1.0F + (p * alpha));
    c[2] = (p + alpha) / (1.0F + (p * alpha));
}

void high_shelf_coeffs(float gain, float w, float c[3])
{
    if (gain >= 1.0F) return pass_through(c);
    const float g = std::max(0.001F, gain);
    const float p = std::sin((0.5F * w) - (0.25F * pi)) / std::sin((0.5F * w) + (0.25F * pi));
    float alpha, beta0, beta1;
    shelf_common(g, alpha, beta0, beta1);
    c[0] = (beta0 + (p * beta1)) / (1.0F + (p * alpha));
    c[1] = (beta1 + (p * beta0)) / (1.0F + (p * alpha));
    c[2] = -(p + alpha) / (1.0F + (p * alpha));
}

// 3-band T60 damping for one line: pick the two first-order sections from the ordering of the
// low / mid / high band decay gains (reference calc_t60_damping_coeffs, src/oalsfxpp.cpp:6922-7010).
void t60_coeffs(float length, float lf_time, float mf_time, float hf_time, float lf_w, float hf_w,
                float lf[3], float hf[3], float& mid)
{
    const float lf_gain = decay_coeff(length, lf_time);
    const float mf_gain = decay_coeff(length, mf_time);
    const float hf_gain = decay_coeff(length, hf_time);

    if (lf_gain < mf_gain) {
        if (mf_gain < hf_gain) {
            low_shelf_coeffs(mf_gain / hf_gain, hf_w, lf);
            highpass_coeffs(lf_gain / mf_gain, lf_w, hf);
            mid = hf_gain;
        } else if (mf_gain > hf_gain) {
            highpass_coeffs(lf_gain / mf_gain, lf_w, lf);
            lowpass_coeffs(hf_gain / mf_gain, hf_w, hf);
            mid = mf_gain;
        } else {
            pass_through(lf);
            highpass_coeffs(lf_gain / mf_gain, lf_w, hf);
            mid = mf_gain;
        }
    } else if (lf_gain > mf_gain) {
        if (mf_gain < hf_gain) {
            const float hg = mf_gain / lf_gain;
            const float lg = mf_gain / hf_gain;
            high_shelf_coeffs(hg, lf_w, lf);
            low_shelf_coeffs(lg, hf_w, hf);
            mid = std::max(lf_gain, hf_gain) / std::max(hg, lg);
        } else if (mf_gain > hf_gain) {
            high_shelf_coeffs(mf_gain / lf_gain, lf_w, lf);
            lowpass_coeffs(hf_gain / mf_gain, hf_w, hf);
            mid = lf_gain;
        } else {
            pass_through(lf);
            high_shelf_coeffs(mf_gain / lf_gain, lf_w, hf);
            mid = lf_gain;
        }
    } else {
        pass_through(lf);
        if (mf_gain < hf_gain) {
            low_shelf_coeffs(mf_gain / hf_gain, hf_w, hf);
            mid = hf_gain;
        } else if (mf_gain > hf_gain) {
            lowpass_coeffs(hf_gain / mf_gain, hf_w, hf);
            mid = mf_gain;
        } else {
            pass_through(hf);
            mid = mf_gain;
        }
    }
}

struct M4 { float m[4][4]; };

// r(row, col) = sum_k a(row,k) b(k,col), summed left to right
M4 mul(const M4& a, const M4& b)
{
    M4 r;
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row)
            r.m[row][col] = (a.m[row][0] * b.m[0][col]) + (a.m[row][1] * b.m[1][col]) + (a.m[row][2] * b.m[2][col]) +
                            (a.m[row][3] * b.m[3][col]);
    return r;
}

M4 mul_transposed(const M4& a, const M4& b)
{
    M4 r;
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row)
            r.m[col][row] = (a.m[row][0] * b.m[0][col]) + (a.m[row][1] * b.m[1][col]) + (a.m[row][2] * b.m[2][col]) +
                            (a.m[row][3] * b.m[3][col]);
    return r;
}

// focus towards the pan vector: Z-focus by its length, then rotate about X and Y
// (reference get_transform_from_vector, src/oalsfxpp.cpp:7236-7282)
M4 transform_from_vector(const float* vec)
{
    const float length = std::sqrt((vec[0] * vec[0]) + (vec[1] * vec[1]) + (vec[2] * vec[2]));
    const float sa = std::sin(std::min(length, 1.0F) * (pi / 4.0F));
    const M4 zfocus = {{
        {1.0F / (1.0F + sa), 0.0F, 0.0F, (sa / (1.0F + sa)) / 1.732050808F},
        {0.0F, std::sqrt((1.0F - sa) / (1.0F + sa)), 0.0F, 0.0F},
        {0.0F, 0.0F, std::sqrt((1.0F - sa) / (1.0F + sa)), 0.0F},
        {(sa / (1.0F + sa)) * 1.732050808F, 0.0F, 0.0F, 1.0F / (1.0F + sa)},
    }};
    float a = std::atan2(vec[1], std::sqrt((vec[0] * vec[0]) + (vec[2] * vec[2])));
    const M4 xrot = {{
        {1.0F, 0.0F, 0.0F, 0.0F},
        {0.0F, 1.0F, 0.0F, 0.0F},
        {0.0F, 0.0F, std::cos(a), std::sin(a)},
        {0.0F, 0.0F, -std::sin(a), std::cos(a)},
    }};
    a = std::atan2(-vec[0], vec[2]);
    const M4 yrot = {{
        {1.0F, 0.0F, 0.0F, 0.0F},
        {0.0F, std::cos(a), 0.0F, std::sin(a)},
        {0.0F, 0.0F, 1.0F, 0.0F},
        {0.0F, -std::sin(a), 0.0F, std::cos(a)},
    }};
    return mul(yrot, mul(xrot, zfocus));
}

const M4 a2b = {{
    {0.866025403785F, 0.866025403785F, 0.866025403785F, 0.866025403785F},
    {0.866025403785F, -0.866025403785F, 0.866025403785F, -0.866025403785F},
    {0.866025403785F, -0.866025403785F, -0.866025403785F, 0.866025403785F},
    {0.866025403785F, 0.866025403785F, -0.866025403785F, -0.866025403785F},
}};

void pan_gains(const DeviceDesc& dev, const float* pan, float gain, float out[4][OALSFX_MAX_CHANNELS])
{
    const M4 transform = mul_transposed(transform_from_vector(pan), a2b);
    for (int i = 0; i < 4; ++i) first_order_gains_foa(dev, transform.m[i], gain, out[i]);
}

} // namespace rv

void derive_reverb(const DeviceDesc& dev, bool is_eax, const EffectProps::Reverb& props, oalsfx_reverb_params& o)
{
    using namespace rv;
    const int frequency = dev.rate;

    // ---- rate-only quantities (reference do_update_device, src/oalsfxpp.cpp:5928-5950) ----
    ring_lengths(frequency, o.ring_len);
    oalsfx_reverb_place_rings(o.ring_len, o.ring_off);
    o.mod_coeff = std::pow(modulation_filter_coeff, modulation_filter_const / frequency);
    const float max_multiplier = 1.0F + line_multiplier;
    o.late_feed_tap = static_cast<int>((R::max_reflections_delay + (early_tap_lengths[3] * max_multiplier)) * frequency);

    // ---- property-dependent quantities (reference do_update, src/oalsfxpp.cpp:5952-6076) ----
    o.is_eax = is_eax ? 1 : 0;

    const float hf_scale = props.hf_reference_ / frequency;
    const float gain_hf = std::max(props.gain_hf_, 0.001F);
    design_biquad(FilterKind::high_shelf, gain_hf, hf_scale, rcp_q_from_slope(gain_hf, 1.0F), o.lp);
    const float lf_scale = props.lf_reference_ / frequency;
    const float gain_lf = std::max(props.gain_lf_, 0.001F);
    design_biquad(FilterKind::low_shelf, gain_lf, lf_scale, rcp_q_from_slope(gain_lf, 1.0F), o.hp);

    // main delay taps (update_delay_line, src/oalsfxpp.cpp:7045-7077)
    const float multiplier = 1.0F + (props.density_ * line_multiplier);
    for (int i = 0; i < 4; ++i) {
        float length = props.reflections_delay_ + (early_tap_lengths[i] * multiplier);
        o.early_tap[i] = static_cast<int>(length * frequency);
        length = early_tap_lengths[i] * multiplier;
        o.early_tap_coeff[i] = decay_coeff(length, props.decay_time_);
        length = props.late_reverb_delay_ + (late_line_lengths[i] - late_line_lengths[0]) * 0.25F * multiplier;
        o.late_tap[i] = o.late_feed_tap + static_cast<int>(length * frequency);
    }

    o.ap_feed_coeff = std::sqrt(0.5F) * std::pow(props.diffusion_, 2.0F);

    // early lines (update_early_lines, src/oalsfxpp.cpp:7080-7106)
    for (int i = 0; i < 4; ++i) {
        float length = early_allpass_lengths[i] * multiplier;
        o.early_ap_off[i] = static_cast<int>(length * frequency);
        length = early_line_lengths[i] * multiplier;
        o.early_line_off[i] = static_cast<int>(length * frequency);
        o.early_line_coeff[i] = decay_coeff(length, props.decay_time_);
    }

    // scattering matrix (calc_matrix_coeffs, src/oalsfxpp.cpp:6645-6659)
    {
        const float n = std::sqrt(3.0F);
        const float t = props.diffusion_ * std::atan(n);
        o.mix_x = std::cos(t);
        o.mix_y = std::sin(t) / n;
    }

    float hf_ratio = props.decay_hf_ratio_;
    if (props.decay_hf_limit_ && props.air_absorption_gain_hf_ < 1.0F)
        hf_ratio = limited_hf_ratio(hf_ratio, props.air_absorption_gain_hf_, props.decay_time_);
    const float lf_decay_time = clamp(props.decay_time_ * props.decay_lf_ratio_, R::min_decay_time, R::max_decay_time);
    const float hf_decay_time = clamp(props.decay_time_ * hf_ratio, R::min_decay_time, R::max_decay_time);

    // modulator (update_modulator, src/oalsfxpp.cpp:7015-7042); the index rescale to the new range is
    // state and happens in the process path when it sees a new update_seq
    o.mod_range = std::max(static_cast<int>(props.modulation_time_ * frequency), 1);
    o.mod_depth = props.modulation_depth_ * modulation_depth_coeff * props.modulation_time_ / 2.0F * frequency;

    // late lines (update_late_lines, src/oalsfxpp.cpp:7109-7196)
    {
        const float lf_w = tau * lf_scale;
        const float hf_w = tau * hf_scale;
        const float mf_decay_time = props.decay_time_;
        const float ap_mean = (late_allpass_lengths[0] + late_allpass_lengths[1] + late_allpass_lengths[2] + late_allpass_lengths[3]) / 4.0F;

        float length = (late_line_lengths[0] + late_line_lengths[1] + late_line_lengths[2] + late_line_lengths[3]) / 4.0F * multiplier;
        length = lerp(length, props.echo_time_, props.echo_depth_);
        length += (late_allpass_lengths[0] + late_allpass_lengths[1] + late_allpass_lengths[2] + late_allpass_lengths[3]) / 4.0F * multiplier;

        const float band_weights[3] = {lf_w, hf_w - lf_w, tau - hf_w};
        o.density_gain = density_gain(decay_coeff(
            length, ((band_weights[0] * lf_decay_time) + (band_weights[1] * mf_decay_time) + (band_weights[2] * hf_decay_time)) / tau));

        for (int i = 0; i < 4; ++i) {
            length = late_allpass_lengths[i] * multiplier;
            o.late_ap_off[i] = static_cast<int>(length * frequency);
            length = lerp(late_line_lengths[i] * multiplier, props.echo_time_, props.echo_depth_);
            o.late_line_off[i] = static_cast<int>(length * frequency);
            length += lerp(late_allpass_lengths[i], ap_mean, props.diffusion_) * multiplier;
            t60_coeffs(length, lf_decay_time, mf_decay_time, hf_decay_time, lf_w, hf_w, o.t60_lf[i], o.t60_hf[i], o.t60_mid[i]);
        }
    }

    // 3D panning of the early and late outputs (update_3d_panning, src/oalsfxpp.cpp:7307-7350)
    pan_gains(dev, props.reflections_pan_.data(), props.gain_ * props.reflections_gain_, o.early_pan);
    pan_gains(dev, props.late_reverb_pan_.data(), props.gain_ * props.late_reverb_gain_, o.late_pan);
}

} // namespace

int ring_floats_for(int type, int rate)
{
    switch (type) {
    case OALSFX_CHORUS: return 2 * next_power_of_2(static_cast<int>(EffectProps::Chorus::max_delay * 2.0F * rate) + 1);
    case OALSFX_FLANGER: return 2 * next_power_of_2(static_cast<int>(EffectProps::Flanger::max_delay * 2.0F * rate) + 1);
    case OALSFX_ECHO: {
        int maxlen = static_cast<int>(EffectProps::Echo::max_delay * rate) + 1;
        maxlen += static_cast<int>(EffectProps::Echo::max_lr_delay * rate) + 1;
        return next_power_of_2(maxlen);
    }
    case OALSFX_REVERB:
    case OALSFX_EAX_REVERB: {
        int len[5];
        rv::ring_lengths(rate, len);
        return 4 * (len[0] + len[1] + len[2] + len[3] + len[4]);
    }
    default: return 0;
    }
}

void derive_slot(const DeviceDesc& dev, const Effect& effect, oalsfx_slot_params& p)
{
    const uint32_t seq = p.update_seq;
    std::memset(&p, 0, sizeof(p));
    p.update_seq = seq;
    p.type = static_cast<int>(effect.type_);
    switch (effect.type_) {
    case EffectType::chorus: derive_moddelay(dev, effect.props_.chorus_, EffectProps::Chorus::max_delay, p.u.moddelay); break;
    case EffectType::flanger: derive_moddelay(dev, effect.props_.flanger_, EffectProps::Flanger::max_delay, p.u.moddelay); break;
    case EffectType::compressor: derive_compressor(dev, effect.props_.compressor_, p.u.compressor); break;
    case EffectType::dedicated_dialog:
    case EffectType::dedicated_low_frequency: derive_dedicated(dev, effect.type_, effect.props_.dedicated_, p.u.dedicated); break;
    case EffectType::distortion: derive_distortion(dev, effect.props_.distortion_, p.u.distortion); break;
    case EffectType::echo: derive_echo(dev, effect.props_.echo_, p.u.echo); break;
    case EffectType::equalizer: derive_equalizer(dev, effect.props_.equalizer_, p.u.equalizer); break;
    case EffectType::ring_modulator: derive_ringmod(dev, effect.props_.ring_modulator_, p.u.ringmod); break;
    case EffectType::reverb: derive_reverb(dev, false, effect.props_.reverb_, p.u.reverb); break;
    case EffectType::eax_reverb: derive_reverb(dev, true, effect.props_.reverb_, p.u.reverb); break;
    case EffectType::null:
    default: break;
    }
}

void reset_slot_state(int type, oalsfx_slot_state& s)
{
    std::memset(&s, 0, sizeof(s));
    switch (type) {
    case OALSFX_COMPRESSOR: s.u.compressor.gain_control = 1.0F; break; // reference src/oalsfxpp.cpp:4312
    case OALSFX_REVERB:
    case OALSFX_EAX_REVERB: s.u.reverb.mod_range = 1; break;           // reference src/oalsfxpp.cpp:5877
    default: break;
    }
}

// ---------------------------------------------------------------------------------------------
// Source sends (reference calc_non_attn_source_params + calc_panning_and_filters,
// src/oalsfxpp.cpp:3172-3395)
// ---------------------------------------------------------------------------------------------
namespace {

struct SpeakerAngle { bool lfe; float azimuth_deg; };

// azimuth of each *input* channel, in the input channel order (reference channel maps, src/oalsfxpp.cpp:3048-3098)
const SpeakerAngle map_mono[] = {{false, 0.0F}};
const SpeakerAngle map_stereo[] = {{false, -30.0F}, {false, 30.0F}};
const SpeakerAngle map_quad[] = {{false, -45.0F}, {false, 45.0F}, {false, -135.0F}, {false, 135.0F}};
const SpeakerAngle map_51[] = {{false, -30.0F}, {false, 30.0F}, {false, 0.0F}, {true, 0.0F}, {false, -110.0F}, {false, 110.0F}};
const SpeakerAngle map_61[] = {{false, -30.0F}, {false, 30.0F}, {false, 0.0F}, {true, 0.0F}, {false, 180.0F}, {false, -90.0F}, {false, 90.0F}};
const SpeakerAngle map_71[] = {{false, -30.0F}, {false, 30.0F}, {false, 0.0F}, {true, 0.0F}, {false, -150.0F}, {false, 150.0F}, {false, -90.0F}, {false, 90.0F}};

constexpr float deg_to_rad(float x) { return x * (pi / 180.0F); }

void send_filters(float gain_hf_in, float gain_lf_in, int frequency, oalsfx_send_params& s)
{
    // NB the shelf reference frequencies are crossed by name, as in the reference (src/oalsfxpp.cpp:3271-3272)
    const float hf_scale = SendProps::hp_frequency_reference / frequency;
    const float lf_scale = SendProps::lp_frequency_reference / frequency;
    const float gain_hf = std::max(gain_hf_in, 0.001F);
    const float gain_lf = std::max(gain_lf_in, 0.001F);
    s.filter_type = OALSFX_AF_NONE;
    if (gain_hf != 1.0F) s.filter_type |= OALSFX_AF_LOW_PASS;
    if (gain_lf != 1.0F) s.filter_type |= OALSFX_AF_HIGH_PASS;
    design_biquad(FilterKind::high_shelf, gain_hf, hf_scale, rcp_q_from_slope(gain_hf, 1.0F), s.lp);
    design_biquad(FilterKind::low_shelf, gain_lf, lf_scale, rcp_q_from_slope(gain_lf, 1.0F), s.hp);
}

} // namespace

void derive_source(const DeviceDesc& dev, int effect_count, const SendProps& direct, const SendProps aux[OALSFX_MAX_SLOTS],
                   const int slot_types[OALSFX_MAX_SLOTS], oalsfx_source_params& out)
{
    std::memset(&out, 0, sizeof(out));
    out.direct.out_channels = dev.channels;
    for (int i = 0; i < effect_count; ++i) out.aux[i].out_channels = (slot_types[i] == OALSFX_NULL) ? 0 : OALSFX_EFFECT_CHANNELS;

    const float dry_gain = std::min(direct.gain_, max_mix_gain);

    const SpeakerAngle* map = nullptr;
    int in_channels = 0;
    switch (dev.format) {
    case oalsfxpp::ChannelFormat::mono: map = map_mono; in_channels = 1; break;
    case oalsfxpp::ChannelFormat::stereo: map = map_stereo; in_channels = 2; break;
    case oalsfxpp::ChannelFormat::quad: map = map_quad; in_channels = 4; break;
    case oalsfxpp::ChannelFormat::five_point_one: map = map_51; in_channels = 6; break;
    case oalsfxpp::ChannelFormat::six_point_one: map = map_61; in_channels = 7; break;
    case oalsfxpp::ChannelFormat::seven_point_one: map = map_71; in_channels = 8; break;
    // five_point_one_rear is absent from the reference's switch (src/oalsfxpp.cpp:3190-3225): no gains at all
    default: break;
    }

    for (int c = 0; c < in_channels; ++c) {
        if (map[c].lfe) continue; // LFE input: all-zero gains (get_channel_index never finds the LFE output)
        float coeffs[max_ambi_coeffs];
        calc_angle_coeffs(deg_to_rad(map[c].azimuth_deg), deg_to_rad(0.0F), 0.0F, coeffs);
        panning_gains_dry(dev, coeffs, dry_gain, out.direct.gains[c]);
        for (int i = 0; i < effect_count; ++i)
            panning_gains_bf(OALSFX_EFFECT_CHANNELS, coeffs, std::min(aux[i].gain_, max_mix_gain), out.aux[i].gains[c]);
    }

    send_filters(direct.gain_hf_, direct.gain_lf_, dev.rate, out.direct);
    for (int i = 0; i < effect_count; ++i) send_filters(aux[i].gain_hf_, aux[i].gain_lf_, dev.rate, out.aux[i]);
}

// ---------------------------------------------------------------------------------------------
// Instance bookkeeping (reference Api::Impl::initialize, Api::apply_changes; src/oalsfxpp.cpp:2846-2905, 3738-3783)
// ---------------------------------------------------------------------------------------------
void InstanceHost::initialize(int count)
{
    effect_count = count;
    for (int i = 0; i < OALSFX_MAX_SLOTS; ++i) {
        std::memset(&deferred[i], 0, sizeof(Effect));
        std::memset(&active[i], 0, sizeof(Effect));
        deferred[i].set_type_and_defaults(EffectType::null);
        active[i].type_ = EffectType::null;
        slot_changed[i] = i < count;
        slot_retyped[i] = i < count;
        aux_props[i].set_defaults();
        aux_deferred[i].set_defaults();
    }
    direct_props.set_defaults();
    direct_deferred.set_defaults();
    source_changed = true;
}

void InstanceHost::apply_changes()
{
    for (int i = 0; i < effect_count; ++i) {
        deferred[i].normalize();
        if (!Effect::are_equal(deferred[i], active[i])) {
            // EffectSlot::set_effect (src/oalsfxpp.cpp:2688-2709): a new type restarts the effect state
            if (active[i].type_ != deferred[i].type_) slot_retyped[i] = true;
            active[i] = deferred[i];
            slot_changed[i] = true;
        }
    }
    direct_deferred.normalize();
    if (!SendProps::are_equal(direct_deferred, direct_props)) {
        source_changed = true;
        direct_props = direct_deferred;
    }
    // Auxiliary sends: set_send_props writes the *active* props directly and apply_changes only compares
    // them with the (never written) deferred copy (src/oalsfxpp.cpp:3728-3733, 3772-3780).
    for (int i = 0; i < effect_count; ++i) {
        aux_deferred[i].normalize();
        if (!SendProps::are_equal(aux_props[i], aux_deferred[i])) source_changed = true;
    }
}

} // namespace oalsfx_host
