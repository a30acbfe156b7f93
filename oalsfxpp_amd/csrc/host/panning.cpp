// Output-device description, ambisonic panning gains and biquad design.
//
// All of this is update-time math (it runs when properties change, never per
// sample); it produces the gain vectors and filter coefficients that the HIP
// kernels consume.  Arithmetic order follows the reference so the derived floats are
// identical: Device::alu_init_renderer (reference src/oalsfxpp.cpp:2489-2570),
// Panning (src/oalsfxpp.cpp:293-808), FilterState::set_params (src/oalsfxpp.cpp:867-982).
#include <cmath>
#include <cstring>

#include "core.hpp"

namespace oalsfx_host {

using oalsfxpp::ChannelFormat;

int next_power_of_2(int value)
{
    // smallest power of two >= value for value > 0 (reference src/oalsfxpp.cpp:189-205)
    if (value <= 0) return value + 1;
    unsigned v = static_cast<unsigned>(value - 1);
    v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return static_cast<int>(v) + 1;
}

int channel_count_of(ChannelFormat format)
{
    switch (format) {
    case ChannelFormat::mono: return 1;
    case ChannelFormat::stereo: return 2;
    case ChannelFormat::quad: return 4;
    case ChannelFormat::five_point_one:
    case ChannelFormat::five_point_one_rear: return 6;
    case ChannelFormat::six_point_one: return 7;
    case ChannelFormat::seven_point_one: return 8;
    default: return 0;
    }
}

namespace {

// Speaker identities (reference ChannelId, src/oalsfxpp.cpp:71-83).
enum Spk { none, FL, FR, FC, LFE, BL, BR, BC, SL, SR };

struct Decoder { Spk spk; float c[max_ambi_coeffs]; };

// Ambisonic decoder rows per layout (values: reference Panning tables, src/oalsfxpp.cpp:295-477).
const Decoder dec_mono[] = {{FC, {1.0F}}};
const Decoder dec_stereo[] = {
    {FL, {5.00000000E-1F, 2.88675135E-1F, 0.0F, 1.19573156E-1F}},
    {FR, {5.00000000E-1F, -2.88675135E-1F, 0.0F, 1.19573156E-1F}},
};
const Decoder dec_quad[] = {
    {BL, {3.53553391E-1F, 2.04124145E-1F, 0.0F, -2.04124145E-1F}},
    {FL, {3.53553391E-1F, 2.04124145E-1F, 0.0F, 2.04124145E-1F}},
    {FR, {3.53553391E-1F, -2.04124145E-1F, 0.0F, 2.04124145E-1F}},
    {BR, {3.53553391E-1F, -2.04124145E-1F, 0.0F, -2.04124145E-1F}},
};
#define X51_ROWS(L, R)                                                                                                    \
    {L, {3.33001372E-1F, 1.89085671E-1F, 0.0F, -2.00041334E-1F, -2.12309737E-2F, 0.0F, 0.0F, 0.0F, -1.14573483E-2F}},  \
    {FL, {1.47751298E-1F, 1.28994110E-1F, 0.0F, 1.15190495E-1F, 7.44949143E-2F, 0.0F, 0.0F, 0.0F, -6.47739980E-3F}},   \
    {FC, {7.73595729E-2F, 0.0F, 0.0F, 9.71390298E-2F, 0.0F, 0.0F, 0.0F, 0.0F, 5.18625335E-2F}},                        \
    {FR, {1.47751298E-1F, -1.28994110E-1F, 0.0F, 1.15190495E-1F, -7.44949143E-2F, 0.0F, 0.0F, 0.0F, -6.47739980E-3F}}, \
    {R, {3.33001372E-1F, -1.89085671E-1F, 0.0F, -2.00041334E-1F, 2.12309737E-2F, 0.0F, 0.0F, 0.0F, -1.14573483E-2F}},
const Decoder dec_51_side[] = {X51_ROWS(SL, SR)};
const Decoder dec_51_rear[] = {X51_ROWS(BL, BR)};
#undef X51_ROWS
const Decoder dec_61[] = {
    {SL, {2.04462744E-1F, 2.17178497E-1F, 0.0F, -4.39990188E-2F, -2.60787329E-2F, 0.0F, 0.0F, 0.0F, -6.87238843E-2F}},
    {FL, {1.18130342E-1F, 9.34633906E-2F, 0.0F, 1.08553749E-1F, 6.80658795E-2F, 0.0F, 0.0F, 0.0F, 1.08999485E-2F}},
    {FC, {7.73595729E-2F, 0.0F, 0.0F, 9.71390298E-2F, 0.0F, 0.0F, 0.0F, 0.0F, 5.18625335E-2F}},
    {FR, {1.18130342E-1F, -9.34633906E-2F, 0.0F, 1.08553749E-1F, -6.80658795E-2F, 0.0F, 0.0F, 0.0F, 1.08999485E-2F}},
    {SR, {2.04462744E-1F, -2.17178497E-1F, 0.0F, -4.39990188E-2F, 2.60787329E-2F, 0.0F, 0.0F, 0.0F, -6.87238843E-2F}},
    {BC, {2.50001688E-1F, 0.0F, 0.0F, -2.50000094E-1F, 0.0F, 0.0F, 0.0F, 0.0F, 6.05133395E-2F}},
};
// Six rows only (no front-center row) -- as in the reference (src/oalsfxpp.cpp:428).
const Decoder dec_71[] = {
    {BL, {2.04124145E-1F, 1.08880247E-1F, 0.0F, -1.88586120E-1F, -1.29099444E-1F, 0.0F, 0.0F, 0.0F, 7.45355993E-2F, 3.73460789E-2F}},
    {SL, {2.04124145E-1F, 2.17760495E-1F, 0.0F, 0.0F, 0.0F, 0.0F, 0.0F, 0.0F, -1.49071198E-1F, -3.73460789E-2F}},
    {FL, {2.04124145E-1F, 1.08880247E-1F, 0.0F, 1.88586120E-1F, 1.29099444E-1F, 0.0F, 0.0F, 0.0F, 7.45355993E-2F, 3.73460789E-2F}},
    {FR, {2.04124145E-1F, -1.08880247E-1F, 0.0F, 1.88586120E-1F, -1.29099444E-1F, 0.0F, 0.0F, 0.0F, 7.45355993E-2F, -3.73460789E-2F}},
    {SR, {2.04124145E-1F, -2.17760495E-1F, 0.0F, 0.0F, 0.0F, 0.0F, 0.0F, 0.0F, -1.49071198E-1F, 3.73460789E-2F}},
    {BR, {2.04124145E-1F, -1.08880247E-1F, 0.0F, -1.88586120E-1F, 1.29099444E-1F, 0.0F, 0.0F, 0.0F, 7.45355993E-2F, -3.73460789E-2F}},
};

struct Layout { const Spk* order; const Decoder* dec; int dec_rows; int coeff_count; };

// Output channel order per format (reference set_default_wfx_channel_order, src/oalsfxpp.cpp:2422-2487).
const Spk ord_mono[] = {FC, none};
const Spk ord_stereo[] = {FL, FR, none};
const Spk ord_quad[] = {FL, FR, BL, BR, none};
const Spk ord_51[] = {FL, FR, FC, LFE, SL, SR, none};
const Spk ord_51r[] = {FL, FR, FC, LFE, BL, BR, none};
const Spk ord_61[] = {FL, FR, FC, LFE, BC, SL, SR, none};
const Spk ord_71[] = {FL, FR, FC, LFE, BL, BR, SL, SR, none};

bool layout_of(ChannelFormat f, Layout& l)
{
#define ROWS(a) a, static_cast<int>(sizeof(a) / sizeof(a[0]))
    switch (f) {
    case ChannelFormat::mono: l = {ord_mono, ROWS(dec_mono), 1}; return true;
    case ChannelFormat::stereo: l = {ord_stereo, ROWS(dec_stereo), 4}; return true;
    case ChannelFormat::quad: l = {ord_quad, ROWS(dec_quad), 4}; return true;
    case ChannelFormat::five_point_one: l = {ord_51, ROWS(dec_51_side), 9}; return true;
    case ChannelFormat::five_point_one_rear: l = {ord_51r, ROWS(dec_51_rear), 9}; return true;
    case ChannelFormat::six_point_one: l = {ord_61, ROWS(dec_61), 9}; return true;
    case ChannelFormat::seven_point_one: l = {ord_71, ROWS(dec_71), 16}; return true;
    default: return false;
    }
#undef ROWS
}

} // namespace

void DeviceDesc::init(ChannelFormat fmt, int sampling_rate)
{
    std::memset(this, 0, sizeof(*this));
    format = fmt;
    rate = sampling_rate;
    channels = channel_count_of(fmt);
    Layout l{};
    if (!layout_of(fmt, l)) return;

    // Per output channel: LFE gets an all-zero row, a speaker without a decoder row
    // (7.1 front-center) keeps zeros too (reference set_channel_map, src/oalsfxpp.cpp:769-807).
    int n = 0;
    for (; n < OALSFX_MAX_CHANNELS && l.order[n] != none; ++n) {
        if (l.order[n] == LFE) continue;
        for (int r = 0; r < l.dec_rows; ++r) {
            if (l.dec[r].spk != l.order[n]) continue;
            for (int k = 0; k < max_ambi_coeffs; ++k) dry[n][k] = l.dec[r].c[k];
            break;
        }
    }
    channels = n;
    dry_coeff_count = l.coeff_count;
    for (int i = 0; i < channels; ++i)
        for (int k = 0; k < 4; ++k) foa[i][k] = dry[i][k];
}

// ---- direction -> ambisonic coefficients (reference src/oalsfxpp.cpp:483-597) ----
static void calc_direction_coeffs(const float dir[3], float spread, float coeffs[max_ambi_coeffs])
{
    // OpenAL -> ambisonic axes
    const float x = -dir[2];
    const float y = -dir[0];
    const float z = dir[1];

    coeffs[0] = 1.0F;
    coeffs[1] = 1.732050808F * y;
    coeffs[2] = 1.732050808F * z;
    coeffs[3] = 1.732050808F * x;
    coeffs[4] = 3.872983346F * x * y;
    coeffs[5] = 3.872983346F * y * z;
    coeffs[6] = 1.118033989F * ((3.0F * z * z) - 1.0F);
    coeffs[7] = 3.872983346F * x * z;
    coeffs[8] = 1.936491673F * ((x * x) - (y * y));
    coeffs[9] = 2.091650066F * y * ((3.0F * x * x) - (y * y));
    coeffs[10] = 10.246950766F * z * x * y;
    coeffs[11] = 1.620185175F * y * ((5.0F * z * z) - 1.0F);
    coeffs[12] = 1.322875656F * z * ((5.0F * z * z) - 3.0F);
    coeffs[13] = 1.620185175F * x * ((5.0F * z * z) - 1.0F);
    coeffs[14] = 5.123475383F * z * ((x * x) - (y * y));
    coeffs[15] = 2.091650066F * x * ((x * x) - (3.0F * y * y));

    if (spread > 0.0F) {
        // spherical-cap source, loudness compensated (zonal-harmonic scale per order)
        const float ca = std::cos(spread * 0.5F);
        const float scale = std::sqrt(1.0F + (spread / tau));
        const float zh[4] = {
            scale,
            0.5F * (ca + 1.0F) * scale,
            0.5F * (ca + 1.0F) * ca * scale,
            0.125F * (ca + 1.0F) * ((5.0F * ca * ca) - 1.0F) * scale,
        };
        static const int order_of[max_ambi_coeffs] = {0, 1, 1, 1, 2, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3};
        for (int k = 0; k < max_ambi_coeffs; ++k) coeffs[k] *= zh[order_of[k]];
    }
}

void calc_angle_coeffs(float azimuth, float elevation, float spread, float coeffs[max_ambi_coeffs])
{
    const float dir[3] = {
        std::sin(azimuth) * std::cos(elevation),
        std::sin(elevation),
        -std::cos(azimuth) * std::cos(elevation),
    };
    calc_direction_coeffs(dir, spread, coeffs);
}

// compute_panning_gains with the dry output (reference src/oalsfxpp.cpp:643-696)
void panning_gains_dry(const DeviceDesc& dev, const float coeffs[max_ambi_coeffs], float in_gain, float out[OALSFX_MAX_CHANNELS])
{
    if (dev.dry_coeff_count <= 0) {
        panning_gains_bf(dev.channels, coeffs, in_gain, out);
        return;
    }
    for (int i = 0; i < OALSFX_MAX_CHANNELS; ++i) {
        if (i >= dev.channels) { out[i] = 0.0F; continue; }
        float gain = 0.0F;
        for (int j = 0; j < dev.dry_coeff_count; ++j) gain += dev.dry[i][j] * coeffs[j];
        out[i] = clamp(gain, 0.0F, 1.0F) * in_gain;
    }
}

// compute_panning_gains_bf (reference src/oalsfxpp.cpp:698-708)
void panning_gains_bf(int channel_count, const float coeffs[max_ambi_coeffs], float in_gain, float out[OALSFX_MAX_CHANNELS])
{
    for (int i = 0; i < OALSFX_MAX_CHANNELS; ++i) out[i] = (i < channel_count ? coeffs[i] * in_gain : 0.0F);
}

// compute_first_order_gains with the first-order output (reference src/oalsfxpp.cpp:713-755; foa_.coeff_count_ is always 4)
void first_order_gains_foa(const DeviceDesc& dev, const float m[4], float in_gain, float out[OALSFX_MAX_CHANNELS])
{
    for (int i = 0; i < OALSFX_MAX_CHANNELS; ++i) {
        if (i >= dev.channels) { out[i] = 0.0F; continue; }
        float gain = 0.0F;
        for (int j = 0; j < 4; ++j) gain += dev.foa[i][j] * m[j];
        out[i] = clamp(gain, 0.0F, 1.0F) * in_gain;
    }
}

// compute_ambient_gains with the dry output (reference src/oalsfxpp.cpp:600-639)
void ambient_gains_dry(const DeviceDesc& dev, float in_gain, float out[OALSFX_MAX_CHANNELS])
{
    for (int i = 0; i < OALSFX_MAX_CHANNELS; ++i) {
        if (dev.dry_coeff_count > 0) out[i] = (i < dev.channels ? dev.dry[i][0] * 1.414213562F * in_gain : 0.0F);
        else out[i] = (i == 0 ? 1.414213562F * in_gain : 0.0F);
    }
}

// ---- RBJ cookbook biquads (reference FilterState::set_params, src/oalsfxpp.cpp:867-982) ----
void design_biquad(FilterKind kind, float gain, float freq_mult, float rcp_q, oalsfx_biquad_t& out)
{
    const float w0 = tau * freq_mult;
    const float sin_w0 = std::sin(w0);
    const float cos_w0 = std::cos(w0);
    const float alpha = sin_w0 / 2.0F * rcp_q;
    float a[3] = {1.0F, 0.0F, 0.0F};
    float b[3] = {1.0F, 0.0F, 0.0F};

    switch (kind) {
    case FilterKind::high_shelf: {
        const float sg = 2.0F * std::sqrt(gain) * alpha;
        b[0] = gain * ((gain + 1.0F) + ((gain - 1.0F) * cos_w0) + sg);
        b[1] = -2.0F * gain * ((gain - 1.0F) + ((gain + 1.0F) * cos_w0));
        b[2] = gain * ((gain + 1.0F) + ((gain - 1.0F) * cos_w0) - sg);
        a[0] = (gain + 1.0F) - ((gain - 1.0F) * cos_w0) + sg;
        a[1] = 2.0F * ((gain - 1.0F) - ((gain + 1.0F) * cos_w0));
        a[2] = (gain + 1.0F) - ((gain - 1.0F) * cos_w0) - sg;
        break;
    }
    case FilterKind::low_shelf: {
        const float sg = 2.0F * std::sqrt(gain) * alpha;
        b[0] = gain * ((gain + 1.0F) - ((gain - 1.0F) * cos_w0) + sg);
        b[1] = 2.0F * gain * ((gain - 1.0F) - ((gain + 1.0F) * cos_w0));
        b[2] = gain * ((gain + 1.0F) - ((gain - 1.0F) * cos_w0) - sg);
        a[0] = (gain + 1.0F) + ((gain - 1.0F) * cos_w0) + sg;
        a[1] = -2.0F * ((gain - 1.0F) + ((gain + 1.0F) * cos_w0));
        a[2] = (gain + 1.0F) + ((gain - 1.0F) * cos_w0) - sg;
        break;
    }
    case FilterKind::peaking: {
        const float sq = std::sqrt(gain);
        b[0] = 1.0F + (alpha * sq);
        b[1] = -2.0F * cos_w0;
        b[2] = 1.0F - (alpha * sq);
        a[0] = 1.0F + (alpha / sq);
        a[1] = -2.0F * cos_w0;
        a[2] = 1.0F - (alpha / sq);
        break;
    }
    case FilterKind::low_pass:
        b[0] = (1.0F - cos_w0) / 2.0F;
        b[1] = 1.0F - cos_w0;
        b[2] = (1.0F - cos_w0) / 2.0F;
        a[0] = 1.0F + alpha;
        a[1] = -2.0F * cos_w0;
        a[2] = 1.0F - alpha;
        break;
    case FilterKind::high_pass:
        b[0] = (1.0F + cos_w0) / 2.0F;
        b[1] = -(1.0F + cos_w0);
        b[2] = (1.0F + cos_w0) / 2.0F;
        a[0] = 1.0F + alpha;
        a[1] = -2.0F * cos_w0;
        a[2] = 1.0F - alpha;
        break;
    case FilterKind::band_pass:
        b[0] = alpha;
        b[1] = 0;
        b[2] = -alpha;
        a[0] = 1.0F + alpha;
        a[1] = -2.0F * cos_w0;
        a[2] = 1.0F - alpha;
        break;
    }

    out.a1 = a[1] / a[0];
    out.a2 = a[2] / a[0];
    out.b0 = b[0] / a[0];
    out.b1 = b[1] / a[0];
    out.b2 = b[2] / a[0];
}

float rcp_q_from_slope(float gain, float slope)
{
    return std::sqrt((gain + (1.0F / gain)) * ((1.0F / slope) - 1.0F) + 2.0F);
}

float rcp_q_from_bandwidth(float freq_mult, float bandwidth)
{
    const float w0 = tau * freq_mult;
    return 2.0F * std::sinh(std::log(2.0F) / 2.0F * bandwidth * w0 / std::sin(w0));
}

} // namespace oalsfx_host
