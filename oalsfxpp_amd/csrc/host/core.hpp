// Host-side (CPU) half of the backend: the parameter-update path.
//
// BASELINE north_star keeps "the parameter-update path and the oalsfxpp.h surface"
// on the host; this directory is that path, re-stated from the reference's
// behaviour so that it emits the flat descriptors of include/oalsfx_desc.h which the
// HIP kernels consume.  Nothing here touches the GPU.
#ifndef OALSFX_HOST_CORE_HPP
#define OALSFX_HOST_CORE_HPP

#include <algorithm>

#include "oalsfx_desc.h"
#include "oalsfxpp.h"

namespace oalsfx_host {

constexpr int min_sampling_rate = 8000;      // reference src/oalsfxpp.cpp:51
constexpr int max_sampling_rate = 8000000;   // reference src/oalsfxpp.cpp:52
constexpr float max_mix_gain = 16.0F;        // reference src/oalsfxpp.cpp:54
constexpr int max_ambi_coeffs = 16;          // reference src/oalsfxpp.cpp:61-62

constexpr float pi = 3.14159265358979323846F;
constexpr float pi_2 = 1.57079632679489661923F;
constexpr float tau = 6.28318530717958647692F;

template <typename T>
inline T clamp(const T v, const T lo, const T hi)
{
    // same expression as the reference's Math::clamp (src/oalsfxpp.cpp:163-169), NaN behaviour included
    return std::min(hi, std::max(lo, v));
}

inline float lerp(const float a, const float b, const float mu) { return a + ((b - a) * mu); }

int next_power_of_2(int value);

int channel_count_of(oalsfxpp::ChannelFormat format);

// ---- output device description (reference Device, src/oalsfxpp.cpp:2378-2625) ----
struct DeviceDesc {
    oalsfxpp::ChannelFormat format;
    int rate;
    int channels;
    int dry_coeff_count;                                   // dry_.coeff_count_
    float dry[OALSFX_MAX_CHANNELS][max_ambi_coeffs];       // dry_.ambi_.coeffs_
    float foa[OALSFX_MAX_CHANNELS][max_ambi_coeffs];       // foa_.ambi_.coeffs_ (first 4 used)

    void init(oalsfxpp::ChannelFormat format, int rate);
};

// ---- panning helpers (reference Panning, src/oalsfxpp.cpp:293-808) ----
void calc_angle_coeffs(float azimuth, float elevation, float spread, float coeffs[max_ambi_coeffs]);
void panning_gains_dry(const DeviceDesc& dev, const float coeffs[max_ambi_coeffs], float in_gain, float out[OALSFX_MAX_CHANNELS]);
void panning_gains_bf(int channel_count, const float coeffs[max_ambi_coeffs], float in_gain, float out[OALSFX_MAX_CHANNELS]);
void first_order_gains_foa(const DeviceDesc& dev, const float matrix_row[4], float in_gain, float out[OALSFX_MAX_CHANNELS]);
void ambient_gains_dry(const DeviceDesc& dev, float in_gain, float out[OALSFX_MAX_CHANNELS]);

// ---- biquad design (reference FilterState::set_params, src/oalsfxpp.cpp:867-982) ----
enum class FilterKind { high_shelf, low_shelf, peaking, low_pass, high_pass, band_pass };
void design_biquad(FilterKind kind, float gain, float freq_mult, float rcp_q, oalsfx_biquad_t& out);
float rcp_q_from_slope(float gain, float slope);
float rcp_q_from_bandwidth(float freq_mult, float bandwidth);

// ---- derivation of the descriptors ----
// Ring footprint (floats) an effect of `type` needs at `rate` (0 for ring-less effects).
int ring_floats_for(int type, int rate);

// What EffectState::update_device + EffectState::update compute for one slot.
// `p.update_seq` is left untouched; `p.type` is set.
void derive_slot(const DeviceDesc& dev, const oalsfxpp::Effect& effect, oalsfx_slot_params& p);

// Fresh process-path state for a newly created effect of `type`
// (reference do_construct of each EffectState).
void reset_slot_state(int type, oalsfx_slot_state& s);

// What calc_non_attn_source_params / calc_panning_and_filters compute.
void derive_source(const DeviceDesc& dev, int effect_count, const oalsfxpp::SendProps& direct,
                   const oalsfxpp::SendProps aux[OALSFX_MAX_SLOTS], const int slot_types[OALSFX_MAX_SLOTS],
                   oalsfx_source_params& out);

// ---- one instance's API-visible bookkeeping (reference Api + Api::Impl minus the sample path) ----
struct InstanceHost {
    int effect_count = 0;
    oalsfxpp::Effect deferred[OALSFX_MAX_SLOTS];
    oalsfxpp::Effect active[OALSFX_MAX_SLOTS];
    bool slot_changed[OALSFX_MAX_SLOTS] = {};   // EffectSlot::is_props_changed_
    bool slot_retyped[OALSFX_MAX_SLOTS] = {};   // type changed since the last sync: state and rings restart from zero
    oalsfxpp::SendProps direct_props, direct_deferred;
    oalsfxpp::SendProps aux_props[OALSFX_MAX_SLOTS], aux_deferred[OALSFX_MAX_SLOTS];
    bool source_changed = false;                // Source::are_props_changed_

    void initialize(int effect_count);
    void apply_changes();
};

} // namespace oalsfx_host

#endif
