// Property structs of the public header: defaults, clamping, comparison, presets.
// Behaviour follows reference src/oalsfxpp.cpp:1158-1930 (EffectProps::*, Effect,
// SendProps) and :1938-2191 (preset values, kept as generated data in presets_data.inc).
#include "core.hpp"

namespace oalsfxpp {

// Field lists: F(name) for scalar fields declared as `name_` with limits min_/max_/default_<name>.
#define CHORUS_FIELDS(F) F(waveform) F(phase) F(rate) F(depth) F(feedback) F(delay)
#define DEDICATED_FIELDS(F) F(gain)
#define DISTORTION_FIELDS(F) F(edge) F(gain) F(low_pass_cutoff) F(eq_center) F(eq_bandwidth)
#define ECHO_FIELDS(F) F(delay) F(lr_delay) F(damping) F(feedback) F(spread)
#define EQUALIZER_FIELDS(F)                                                                          \
    F(low_cutoff) F(low_gain) F(mid1_center) F(mid1_gain) F(mid1_width) F(mid2_center) F(mid2_gain) \
    F(mid2_width) F(high_cutoff) F(high_gain)
#define RINGMOD_FIELDS(F) F(frequency) F(high_pass_cutoff) F(waveform)
#define SEND_FIELDS(F) F(gain) F(gain_hf) F(gain_lf)
// Reverb scalars before, between and after the two pan vectors (declaration order matters for are_equal only
// in the sense that every field is compared).
#define REVERB_SCALARS(F)                                                                                   \
    F(density) F(diffusion) F(gain) F(gain_hf) F(gain_lf) F(decay_time) F(decay_hf_ratio) F(decay_lf_ratio) \
    F(reflections_gain) F(reflections_delay) F(late_reverb_gain) F(late_reverb_delay) F(echo_time)          \
    F(echo_depth) F(modulation_time) F(modulation_depth) F(air_absorption_gain_hf) F(hf_reference)          \
    F(lf_reference) F(room_rolloff_factor) F(decay_hf_limit)

#define F_DEFAULT(n) n##_ = default_##n;
#define F_CLAMP(n) n##_ = oalsfx_host::clamp(n##_, min_##n, max_##n);
#define F_EQUAL(n) &&a.n##_ == b.n##_

#define DEFINE_OPS(S, FIELDS)                                                   \
    void S::set_defaults() { FIELDS(F_DEFAULT) }                               \
    void S::normalize() { FIELDS(F_CLAMP) }                                    \
    bool S::are_equal(const S& a, const S& b) { return true FIELDS(F_EQUAL); }

DEFINE_OPS(EffectProps::Chorus, CHORUS_FIELDS)
DEFINE_OPS(EffectProps::Flanger, CHORUS_FIELDS)
DEFINE_OPS(EffectProps::Dedicated, DEDICATED_FIELDS)
DEFINE_OPS(EffectProps::Distortion, DISTORTION_FIELDS)
DEFINE_OPS(EffectProps::Echo, ECHO_FIELDS)
DEFINE_OPS(EffectProps::Equalizer, EQUALIZER_FIELDS)
DEFINE_OPS(EffectProps::RingModulator, RINGMOD_FIELDS)
DEFINE_OPS(SendProps, SEND_FIELDS)

// The compressor has a single bool and nothing to clamp (reference src/oalsfxpp.cpp:1441-1457).
void EffectProps::Compressor::set_defaults() { on_off_ = default_on_off; }
void EffectProps::Compressor::normalize() {}
bool EffectProps::Compressor::are_equal(const Compressor& a, const Compressor& b) { return a.on_off_ == b.on_off_; }

void EffectProps::Reverb::set_defaults()
{
    REVERB_SCALARS(F_DEFAULT)
    reflections_pan_.fill(default_reflections_pan_xyz);
    late_reverb_pan_.fill(default_late_reverb_pan_xyz);
}

void EffectProps::Reverb::normalize()
{
    REVERB_SCALARS(F_CLAMP)
    for (auto& v : reflections_pan_) v = oalsfx_host::clamp(v, min_reflections_pan_xyz, max_reflections_pan_xyz);
    for (auto& v : late_reverb_pan_) v = oalsfx_host::clamp(v, min_late_reverb_pan_xyz, max_late_reverb_pan_xyz);
}

bool EffectProps::Reverb::are_equal(const Reverb& a, const Reverb& b)
{
    return a.reflections_pan_ == b.reflections_pan_ && a.late_reverb_pan_ == b.late_reverb_pan_ REVERB_SCALARS(F_EQUAL);
}

// ---- Effect: dispatch on the type tag (reference src/oalsfxpp.cpp:1726-1878) ----
namespace {

enum class Op { defaults, normalize };

void for_type(Effect& e, Op op)
{
#define APPLY(member)                                   \
    if (op == Op::defaults) e.props_.member.set_defaults(); \
    else e.props_.member.normalize();                   \
    break;
    switch (e.type_) {
    case EffectType::chorus: APPLY(chorus_)
    case EffectType::compressor: APPLY(compressor_)
    case EffectType::dedicated_dialog:
    case EffectType::dedicated_low_frequency: APPLY(dedicated_)
    case EffectType::distortion: APPLY(distortion_)
    case EffectType::echo: APPLY(echo_)
    case EffectType::equalizer: APPLY(equalizer_)
    case EffectType::flanger: APPLY(flanger_)
    case EffectType::reverb:
    case EffectType::eax_reverb: APPLY(reverb_)
    case EffectType::ring_modulator: APPLY(ring_modulator_)
    case EffectType::null:
    default: break;
    }
#undef APPLY
}

} // namespace

void Effect::set_defaults() { for_type(*this, Op::defaults); }

void Effect::set_type_and_defaults(const EffectType effect_type)
{
    type_ = effect_type;
    set_defaults();
}

void Effect::normalize() { for_type(*this, Op::normalize); }

bool Effect::are_equal(const Effect& a, const Effect& b)
{
    if (a.type_ != b.type_) return false;
    switch (a.type_) {
    case EffectType::null: return true;
    case EffectType::chorus: return EffectProps::Chorus::are_equal(a.props_.chorus_, b.props_.chorus_);
    case EffectType::compressor: return EffectProps::Compressor::are_equal(a.props_.compressor_, b.props_.compressor_);
    case EffectType::dedicated_dialog:
    case EffectType::dedicated_low_frequency: return EffectProps::Dedicated::are_equal(a.props_.dedicated_, b.props_.dedicated_);
    case EffectType::distortion: return EffectProps::Distortion::are_equal(a.props_.distortion_, b.props_.distortion_);
    case EffectType::echo: return EffectProps::Echo::are_equal(a.props_.echo_, b.props_.echo_);
    case EffectType::equalizer: return EffectProps::Equalizer::are_equal(a.props_.equalizer_, b.props_.equalizer_);
    case EffectType::flanger: return EffectProps::Flanger::are_equal(a.props_.flanger_, b.props_.flanger_);
    case EffectType::reverb:
    case EffectType::eax_reverb: return EffectProps::Reverb::are_equal(a.props_.reverb_, b.props_.reverb_);
    case EffectType::ring_modulator: return EffectProps::RingModulator::are_equal(a.props_.ring_modulator_, b.props_.ring_modulator_);
    default: return false;
    }
}

// ---- presets: values are data generated from the compiled reference (oracle/gen_presets.py) ----
#define OALSFX_PRESET_DATA(group, name, ...) const EffectProps::Reverb ReverbPresets::group::name = {__VA_ARGS__};
#include "presets_data.inc"
#undef OALSFX_PRESET_DATA


// Out-of-class definitions so that C++14 callers may ODR-use the limit constants
// (the reference provides them too, src/oalsfxpp.cpp:1158-1404).
#define F_DEFINE_IN(S, n)                             \
    constexpr decltype(S::min_##n) S::min_##n;        \
    constexpr decltype(S::max_##n) S::max_##n;        \
    constexpr decltype(S::default_##n) S::default_##n;
#define F_DEF_CHORUS(n) F_DEFINE_IN(EffectProps::Chorus, n)
#define F_DEF_FLANGER(n) F_DEFINE_IN(EffectProps::Flanger, n)
#define F_DEF_DEDICATED(n) F_DEFINE_IN(EffectProps::Dedicated, n)
#define F_DEF_DISTORTION(n) F_DEFINE_IN(EffectProps::Distortion, n)
#define F_DEF_ECHO(n) F_DEFINE_IN(EffectProps::Echo, n)
#define F_DEF_EQUALIZER(n) F_DEFINE_IN(EffectProps::Equalizer, n)
#define F_DEF_RINGMOD(n) F_DEFINE_IN(EffectProps::RingModulator, n)
#define F_DEF_REVERB(n) F_DEFINE_IN(EffectProps::Reverb, n)
#define F_DEF_SEND(n) F_DEFINE_IN(SendProps, n)
CHORUS_FIELDS(F_DEF_CHORUS)
CHORUS_FIELDS(F_DEF_FLANGER)
DEDICATED_FIELDS(F_DEF_DEDICATED)
DISTORTION_FIELDS(F_DEF_DISTORTION)
ECHO_FIELDS(F_DEF_ECHO)
EQUALIZER_FIELDS(F_DEF_EQUALIZER)
RINGMOD_FIELDS(F_DEF_RINGMOD)
REVERB_SCALARS(F_DEF_REVERB)
F_DEF_REVERB(reflections_pan_xyz)
F_DEF_REVERB(late_reverb_pan_xyz)
SEND_FIELDS(F_DEF_SEND)
F_DEFINE_IN(EffectProps::Compressor, on_off)
constexpr int EffectProps::Chorus::waveform_sinusoid;
constexpr int EffectProps::Chorus::waveform_triangle;
constexpr int EffectProps::Flanger::waveform_sinusoid;
constexpr int EffectProps::Flanger::waveform_triangle;
constexpr int EffectProps::RingModulator::waveform_sinusoid;
constexpr int EffectProps::RingModulator::waveform_sawtooth;
constexpr int EffectProps::RingModulator::waveform_square;
constexpr float SendProps::lp_frequency_reference;
constexpr float SendProps::hp_frequency_reference;

static_assert(sizeof(EffectProps) == 108, "EffectProps layout must match the reference (SURVEY 8b)");
static_assert(sizeof(Effect) == 112, "Effect layout must match the reference (SURVEY 8b)");
static_assert(sizeof(SendProps) == 12, "SendProps layout must match the reference (SURVEY 8b)");

} // namespace oalsfxpp
