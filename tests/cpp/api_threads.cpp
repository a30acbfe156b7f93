// The reference's threading contract: an Api object is not thread-safe, distinct Api objects are independent and may
// be driven from different threads at the same time.  Eight Api objects with different effects are run twice -- one after the
// other on the main thread, then four threads with two objects each, all mixing at once -- and every output must be bit-identical
// between the two runs.  Linked against liboalsfx_hip.so through include/oalsfxpp.h only.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "oalsfxpp.h"

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

static const int kObjects = 8, kBuffers = 14;
static const int kSizes[kBuffers] = {256, 256, 256, 100, 256, 441, 256, 256, 2500, 64, 256, 256, 31, 256};

// one object's whole life: initialize, program, mix kBuffers buffers (with a property change in the middle), collect the output
static bool run_object(int id, std::vector<float>& out, std::string& error)
{
    Api api;
    const int slots = 1 + id % 2;
    if (!api.initialize(id % 3 == 2 ? ChannelFormat::mono : ChannelFormat::stereo, 44100 + 3900 * (id % 2), slots)) { error = api.get_error_message(); return false; }
    const int ch = api.get_channel_count();
    static const EffectType types[kObjects] = {EffectType::eax_reverb, EffectType::echo, EffectType::reverb, EffectType::distortion,
                                              EffectType::eax_reverb, EffectType::chorus, EffectType::equalizer, EffectType::eax_reverb};
    Effect e;
    e.set_type_and_defaults(types[id]);
    if (id == 4) e.props_.reverb_ = ReverbPresets::Misc::small_water_room;
    if (id == 7) e.props_.reverb_ = ReverbPresets::Default::psychotic;
    api.set_effect(0, e);
    if (slots == 2 && !api.set_effect_type(1, EffectType::flanger)) { error = api.get_error_message(); return false; }
    if (!api.apply_changes()) { error = api.get_error_message(); return false; }
    for (int k = 0; k < kBuffers; ++k) {
        if (k == 6) {
            // a change while streaming: another preset (cross-fade) or another effect type (state re-created)
            Effect c;
            if (types[id] == EffectType::eax_reverb) { c.set_type_and_defaults(EffectType::eax_reverb); c.props_.reverb_ = ReverbPresets::Default::cave; }
            else c.set_type_and_defaults(EffectType::ring_modulator);
            api.set_effect(0, c);
            api.set_send_props(-1, SendProps{0.9F, 0.7F, 1.0F});
            if (!api.apply_changes()) { error = api.get_error_message(); return false; }
        }
        const int frames = kSizes[k];
        std::vector<float> src(static_cast<size_t>(frames) * ch), dst(src.size());
        synth(static_cast<uint32_t>(300 + id), static_cast<uint32_t>(k), static_cast<int>(src.size()), src.data());
        if (!api.mix(frames, src.data(), dst.data())) { error = api.get_error_message(); return false; }
        out.insert(out.end(), dst.begin(), dst.end());
    }
    return true;
}

int main()
{
    std::vector<std::vector<float>> serial(kObjects), threaded(kObjects);
    std::vector<std::string> errors(kObjects);
    for (int id = 0; id < kObjects; ++id)
        if (!run_object(id, serial[id], errors[id])) { std::fprintf(stderr, "serial run, object %d: %s\n", id, errors[id].c_str()); return 2; }
    std::vector<int> ok(kObjects, 0);
    std::vector<std::thread> threads;
    for (int t = 0; t < 4; ++t)
        threads.emplace_back([&, t]() {
            for (int id = t; id < kObjects; id += 4) ok[id] = run_object(id, threaded[id], errors[id]) ? 1 : 0;
        });
    for (auto& th : threads) th.join();
    for (int id = 0; id < kObjects; ++id) {
        if (!ok[id]) { std::fprintf(stderr, "threaded run, object %d: %s\n", id, errors[id].c_str()); return 3; }
        if (serial[id].size() != threaded[id].size() || std::memcmp(serial[id].data(), threaded[id].data(), serial[id].size() * sizeof(float)) != 0) {
            std::fprintf(stderr, "object %d: outputs of the threaded run differ from the serial run\n", id);
            return 4;
        }
    }
    std::puts("ok");
    return 0;
}
