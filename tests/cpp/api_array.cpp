// oalsfxpp::ApiArray (include/oalsfxpp_array.h) against the same chains as separate oalsfxpp::Api objects: forty voices, each with
// buffers of its own, effects of several types, a change while streaming and odd call sizes -- every output bit-identical, and the
// error behaviour the reference's (set_effect returns false on success, messages for bad indices and null buffers).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "oalsfxpp_array.h"

using namespace oalsfxpp;

static void synth(uint32_t instance, uint32_t buffer_index, int count, float* out)
{
    uint32_t x = 0x9E3779B9u ^ (instance * 2654435761u) ^ buffer_index;
    if (x == 0) x = 1;
    for (int i = 0; i < count; ++i) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        out[i] = static_cast<float>(x >> 8) * (1.0F / 8388608.0F) - 1.0F;
    }
}

static Effect effect_for(int i)
{
    static const EffectType types[] = {EffectType::eax_reverb, EffectType::echo, EffectType::reverb, EffectType::chorus, EffectType::eax_reverb, EffectType::equalizer};
    Effect e;
    e.set_type_and_defaults(types[i % 6]);
    if (i % 6 == 4) e.props_.reverb_ = ReverbPresets::Default::cave;
    return e;
}

#define CHECK(cond, ...) do { if (!(cond)) { std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); return 1; } } while (0)

int main()
{
    const int n = 40, slots = 2, ch = 2;
    const int sizes[] = {256, 256, 441, 256, 100, 2100, 64, 256};
    ApiArray arr;
    CHECK(!arr.mix(256, static_cast<const float*>(nullptr), nullptr) && std::strcmp(arr.get_error_message(), "Not initialized.") == 0, "uninitialised mix: %s", arr.get_error_message());
    CHECK(arr.initialize(n, ChannelFormat::stereo, 48000, slots), "initialize: %s", arr.get_error_message());
    CHECK(arr.size() == n && arr.get_channel_count() == ch && arr.get_effect_count() == slots, "sizes");
    std::vector<std::unique_ptr<Api>> apis;
    for (int i = 0; i < n; ++i) {
        apis.emplace_back(new Api);
        CHECK(apis[i]->initialize(ChannelFormat::stereo, 48000, slots), "Api::initialize: %s", apis[i]->get_error_message());
        const Effect e = effect_for(i);
        CHECK(!arr.set_effect(i, 0, e) && !apis[i]->set_effect(0, e), "set_effect returns false on success");
        if (i % 3 == 0) { CHECK(arr.set_effect_type(i, 1, EffectType::flanger) && apis[i]->set_effect_type(1, EffectType::flanger), "set_effect_type"); }
        CHECK(apis[i]->apply_changes(), "Api::apply_changes");
    }
    CHECK(arr.apply_changes(), "apply_changes: %s", arr.get_error_message());
    CHECK(!arr.set_effect_type(n, 0, EffectType::echo) && std::strcmp(arr.get_error_message(), "Instance index is out of range.") == 0, "instance range: %s", arr.get_error_message());
    CHECK(!arr.set_effect_type(0, slots, EffectType::echo) && std::strcmp(arr.get_error_message(), "Effect index is out of range.") == 0, "effect range: %s", arr.get_error_message());
    Effect back;
    CHECK(arr.get_effect(4, 0, back) && back.type_ == EffectType::eax_reverb, "get_effect");
    int k = 0;
    for (int frames : sizes) {
        if (k == 3) {
            for (int i = 0; i < n; i += 5) {
                SendProps sp{0.8F, 0.6F, 1.0F};
                Effect e;
                e.set_type_and_defaults(EffectType::eax_reverb);
                e.props_.reverb_ = ReverbPresets::Misc::small_water_room;
                arr.set_effect(i, 0, e); apis[i]->set_effect(0, e);
                CHECK(arr.set_send_props(i, -1, sp) && apis[i]->set_send_props(-1, sp), "set_send_props");
                CHECK(arr.apply_changes(i) && apis[i]->apply_changes(), "apply_changes(i)");
            }
        }
        std::vector<std::vector<float>> src(n), want(n), got(n);
        std::vector<const float*> sp(n);
        std::vector<float*> dp(n);
        for (int i = 0; i < n; ++i) {
            src[i].resize(static_cast<size_t>(frames) * ch); want[i].resize(src[i].size()); got[i].resize(src[i].size());
            synth(700 + i, k, static_cast<int>(src[i].size()), src[i].data());
            CHECK(apis[i]->mix(frames, src[i].data(), want[i].data()), "Api::mix: %s", apis[i]->get_error_message());
            sp[i] = src[i].data(); dp[i] = got[i].data();
        }
        CHECK(arr.mix(frames, sp.data(), dp.data()), "ApiArray::mix: %s", arr.get_error_message());
        for (int i = 0; i < n; ++i)
            CHECK(std::memcmp(want[i].data(), got[i].data(), want[i].size() * sizeof(float)) == 0, "buffer %d (%d frames): instance %d differs", k, frames, i);
        ++k;
    }
    // the contiguous form
    {
        const int frames = 256;
        std::vector<float> src(static_cast<size_t>(n) * frames * ch), got(src.size()), want(src.size());
        for (int i = 0; i < n; ++i) {
            synth(700 + i, k, frames * ch, src.data() + static_cast<size_t>(i) * frames * ch);
            CHECK(apis[i]->mix(frames, src.data() + static_cast<size_t>(i) * frames * ch, want.data() + static_cast<size_t>(i) * frames * ch), "Api::mix");
        }
        CHECK(arr.mix(frames, src.data(), got.data()), "ApiArray::mix (contiguous): %s", arr.get_error_message());
        CHECK(std::memcmp(want.data(), got.data(), want.size() * sizeof(float)) == 0, "contiguous mix differs");
    }
    const float* none = nullptr;
    std::vector<const float*> bad(n, none);
    std::vector<float*> out(n, nullptr);
    CHECK(!arr.mix(64, bad.data(), out.data()) && std::strcmp(arr.get_error_message(), "No source samples.") == 0, "null per-instance source: %s", arr.get_error_message());
    std::printf("ok\n");
    return 0;
}
