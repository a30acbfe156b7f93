// Host-only entry points of the C ABI (include/oalsfx_hip.h, "host-only helpers").
#include <cstring>

#include "core.hpp"
#include "oalsfx_hip.h"

using namespace oalsfx_host;
using oalsfxpp::Effect;
using oalsfxpp::EffectProps;
using oalsfxpp::ReverbPresets;

static_assert(sizeof(oalsfx_effect) == sizeof(Effect), "oalsfx_effect must mirror oalsfxpp::Effect");
static_assert(sizeof(oalsfx_send_props) == sizeof(oalsfxpp::SendProps), "oalsfx_send_props must mirror oalsfxpp::SendProps");

namespace {

struct PresetEntry { const char* name; const EffectProps::Reverb* props; };

#define OALSFX_PRESET(group, name) {#group "::" #name, &ReverbPresets::group::name},
const PresetEntry presets[] = {
#include "oalsfx_preset_names.inc"
};
#undef OALSFX_PRESET

oalsfxpp::SendProps to_send(const oalsfx_send_props& s)
{
    oalsfxpp::SendProps r;
    r.gain_ = s.gain; r.gain_hf_ = s.gain_hf; r.gain_lf_ = s.gain_lf;
    return r;
}

} // namespace

extern "C" {

void oalsfx_host_effect_defaults(int effect_type, oalsfx_effect* out)
{
    Effect e;
    std::memset(&e, 0, sizeof(e));
    e.set_type_and_defaults(static_cast<oalsfxpp::EffectType>(effect_type));
    std::memcpy(out, &e, sizeof(e));
}

void oalsfx_host_effect_normalize(oalsfx_effect* io)
{
    Effect e;
    std::memcpy(&e, io, sizeof(e));
    e.normalize();
    std::memcpy(io, &e, sizeof(e));
}

int oalsfx_host_derive_slot(int channel_format, int sampling_rate, const oalsfx_effect* normalized, oalsfx_slot_params* out)
{
    DeviceDesc dev;
    dev.init(static_cast<oalsfxpp::ChannelFormat>(channel_format), sampling_rate);
    if (dev.channels == 0) return 0;
    Effect e;
    std::memcpy(&e, normalized, sizeof(e));
    derive_slot(dev, e, *out);
    return 1;
}

int oalsfx_host_derive_source(int channel_format, int sampling_rate, int effect_count, const oalsfx_send_props* direct,
                              const oalsfx_send_props* aux, const int* slot_types, oalsfx_source_params* out)
{
    DeviceDesc dev;
    dev.init(static_cast<oalsfxpp::ChannelFormat>(channel_format), sampling_rate);
    if (dev.channels == 0 || effect_count < 1 || effect_count > OALSFX_MAX_SLOTS) return 0;
    oalsfxpp::SendProps a[OALSFX_MAX_SLOTS];
    int types[OALSFX_MAX_SLOTS] = {};
    for (int i = 0; i < effect_count; ++i) { a[i] = to_send(aux[i]); types[i] = slot_types[i]; }
    derive_source(dev, effect_count, to_send(*direct), a, types, *out);
    return 1;
}

int oalsfx_host_ring_floats(int effect_type, int sampling_rate) { return ring_floats_for(effect_type, sampling_rate); }

int oalsfx_host_channel_count(int channel_format) { return channel_count_of(static_cast<oalsfxpp::ChannelFormat>(channel_format)); }

int oalsfx_host_preset_count(void) { return static_cast<int>(sizeof(presets) / sizeof(presets[0])); }

const char* oalsfx_host_preset_name(int index)
{
    if (index < 0 || index >= oalsfx_host_preset_count()) return nullptr;
    return presets[index].name;
}

int oalsfx_host_preset(int index, void* reverb_props_out)
{
    if (index < 0 || index >= oalsfx_host_preset_count()) return 0;
    std::memcpy(reverb_props_out, presets[index].props, sizeof(EffectProps::Reverb));
    return 1;
}

} // extern "C"
