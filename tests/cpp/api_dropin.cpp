// Drop-in check: a caller written against the reference's public header, compiled against include/oalsfxpp.h and
// linked with liboalsfx_hip.so.  Mixes a few buffers through oalsfxpp::Api and writes the output as raw floats; the
// pytest wrapper compares it with the CPU oracle.  Also exercises the error conventions of the facade.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "oalsfxpp.h"

static void synth(uint32_t instance, uint32_t buffer_index, int count, float* out)
{
    uint32_t x = 0x9E3779B9u ^ (instance * 2654435761u) ^ buffer_index;
    if (x == 0) x = 1;
    for (int i = 0; i < count; ++i) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        out[i] = static_cast<float>(x >> 8) * (1.0F / 8388608.0F) - 1.0F;
    }
}

#define CHECK(cond)                                                     \
    if (!(cond)) {                                                      \
        std::fprintf(stderr, "check failed line %d: %s (last message: %s)\n", __LINE__, #cond, api.get_error_message()); \
        return 2;                                                       \
    }

int main(int argc, char** argv)
{
    using namespace oalsfxpp;
    if (argc < 2) return 1;
    Api api;
    CHECK(!api.is_initialized());
    CHECK(api.get_sampling_rate() == 0);
    CHECK(std::strcmp(api.get_error_message(), "Not initialized.") == 0);
    CHECK(!api.initialize(ChannelFormat::none, 48000, 1));
    CHECK(std::strcmp(api.get_error_message(), "Invalid channel format.") == 0);
    CHECK(!api.initialize(ChannelFormat::stereo, 48000, 9));
    CHECK(std::strcmp(api.get_error_message(), "Effect count is out of range.") == 0);
    CHECK(api.initialize(Api::channel_count_to_channel_format(2), 48000, 2));
    CHECK(api.get_channel_count() == 2 && api.get_effect_count() == 2 && api.get_channel_format() == ChannelFormat::stereo);
    CHECK(!api.set_effect_type(5, EffectType::echo));
    CHECK(std::strcmp(api.get_error_message(), "Effect index is out of range.") == 0);

    Effect e;
    e.set_type_and_defaults(EffectType::eax_reverb);
    e.props_.reverb_ = ReverbPresets::Misc::small_water_room;
    CHECK(api.set_effect(0, e) == false);          // the reference's quirk: stores, returns false
    CHECK(api.set_effect_type(1, EffectType::chorus));
    Effect back{};
    CHECK(api.get_deferred_effect(0, back) && back.type_ == EffectType::eax_reverb);
    CHECK(api.get_effect(0, back) && back.type_ == EffectType::null); // not applied yet
    CHECK(api.apply_changes());
    CHECK(api.get_effect(0, back) && Effect::are_equal(back, e));
    CHECK(api.mix(0, nullptr, nullptr));
    CHECK(!api.mix(16, nullptr, nullptr));
    CHECK(std::strcmp(api.get_error_message(), "No source samples.") == 0);

    std::FILE* f = std::fopen(argv[1], "wb");
    CHECK(f != nullptr);
    const int sizes[] = {256, 256, 256, 100, 3000};
    int k = 0;
    for (int frames : sizes) {
        std::vector<float> src(static_cast<size_t>(frames) * 2), dst(src.size());
        synth(77, static_cast<uint32_t>(k++), static_cast<int>(src.size()), src.data());
        if (!api.mix(frames, src.data(), dst.data())) {
            std::fprintf(stderr, "mix failed: %s\n", api.get_error_message());
            return 3;
        }
        std::fwrite(dst.data(), sizeof(float), dst.size(), f);
    }
    std::fclose(f);
    api.uninitialize();
    CHECK(!api.is_initialized());
    std::puts("ok");
    return 0;
}
