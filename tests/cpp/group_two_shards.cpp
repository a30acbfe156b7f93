// The multi-GPU split behind the C ABI (include/oalsfx_hip.h, oalsfx_group_*), rehearsed on one GPU: a group of two shards that both
// name device 0 must produce, bit for bit, what one batch of all the instances produces -- through the host-pointer call (a host thread
// per shard) and through the device-buffer call (queued shard after shard).  BASELINE configs[4] is this with eight ordinals.
// Usage: group_two_shards [n_total] ; prints "ok" and returns 0.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "oalsfx_hip.h"
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

static oalsfx_effect effect_for(int i)
{
    oalsfxpp::Effect e;
    if (i % 5 == 4) e.set_type_and_defaults(oalsfxpp::EffectType::echo);
    else {
        e.set_type_and_defaults(i % 2 ? oalsfxpp::EffectType::eax_reverb : oalsfxpp::EffectType::reverb);
        if (i % 3 == 0) e.props_.reverb_ = oalsfxpp::ReverbPresets::Default::cave;
        if (i % 7 == 0) e.props_.reverb_ = oalsfxpp::ReverbPresets::Misc::small_water_room;
    }
    oalsfx_effect out;
    static_assert(sizeof(out) == sizeof(e), "oalsfx_effect mirrors oalsfxpp::Effect");
    std::memcpy(&out, &e, sizeof(out));
    return out;
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? std::atoi(argv[1]) : 37, ch = 2;
    const int sizes[] = {256, 256, 100, 256, 441, 2500, 64, 256};
    const int devices[2] = {0, 0};
    oalsfx_group* g = oalsfx_group_create(n, devices, 2, static_cast<int>(oalsfxpp::ChannelFormat::stereo), 48000, 1);
    if (!g) { std::fprintf(stderr, "group: %s\n", oalsfx_group_last_error()); return 1; }
    oalsfx_batch* b = oalsfx_batch_create(n, static_cast<int>(oalsfxpp::ChannelFormat::stereo), 48000, 1, 0);
    if (!b) { std::fprintf(stderr, "batch: %s\n", oalsfx_last_error()); return 1; }
    int first1 = 0, count0 = 0, count1 = 0;
    oalsfx_group_shard(g, 0, nullptr, nullptr, &count0);
    oalsfx_group_shard(g, 1, nullptr, &first1, &count1);
    if (count0 + count1 != n || first1 != count0 || (count0 - count1 != 0 && count0 - count1 != 1)) { std::fprintf(stderr, "shards %d + %d of %d\n", count0, count1, n); return 1; }
    std::vector<oalsfx_effect> effects;
    for (int i = 0; i < n; ++i) effects.push_back(effect_for(i));
    // (a range that straddles the shard boundary, then the rest)
    const int mid0 = count0 > 3 ? count0 - 3 : 0, mid1 = count0 + 2 < n ? count0 + 2 : n;
    bool ok = oalsfx_group_set_effect(g, mid0, mid1 - mid0, 0, effects.data() + mid0, sizeof(oalsfx_effect)) &&
              oalsfx_group_set_effect(g, 0, mid0, 0, effects.data(), sizeof(oalsfx_effect)) &&
              oalsfx_group_set_effect(g, mid1, n - mid1, 0, effects.data() + mid1, sizeof(oalsfx_effect)) && oalsfx_group_apply_changes(g, 0, n);
    if (!ok) { std::fprintf(stderr, "group setters: %s\n", oalsfx_group_error(g)); return 1; }
    ok = oalsfx_batch_set_effect(b, 0, n, 0, effects.data(), sizeof(oalsfx_effect)) && oalsfx_batch_apply_changes(b, 0, n);
    if (!ok) { std::fprintf(stderr, "batch setters: %s\n", oalsfx_batch_error(b)); return 1; }
    int k = 0;
    for (int frames : sizes) {
        std::vector<float> src(static_cast<size_t>(n) * frames * ch), want(src.size()), got(src.size());
        for (int i = 0; i < n; ++i) synth(500 + i, k, frames * ch, src.data() + static_cast<size_t>(i) * frames * ch);
        if (k == 4) {
            // a change while streaming, on both sides of the boundary
            oalsfx_send_props sp{0.8F, 0.5F, 1.0F};
            ok = oalsfx_group_set_send_props(g, count0 - 1, 2, -1, &sp) && oalsfx_group_apply_changes(g, 0, n) &&
                 oalsfx_batch_set_send_props(b, count0 - 1, 2, -1, &sp) && oalsfx_batch_apply_changes(b, 0, n);
            if (!ok) { std::fprintf(stderr, "send props: %s / %s\n", oalsfx_group_error(g), oalsfx_batch_error(b)); return 1; }
        }
        if (!oalsfx_batch_mix(b, frames, src.data(), want.data())) { std::fprintf(stderr, "batch mix: %s\n", oalsfx_batch_error(b)); return 1; }
        if (!oalsfx_group_mix(g, frames, src.data(), got.data())) { std::fprintf(stderr, "group mix: %s\n", oalsfx_group_error(g)); return 1; }
        if (std::memcmp(want.data(), got.data(), want.size() * sizeof(float)) != 0) {
            for (size_t j = 0; j < want.size(); ++j)
                if (std::memcmp(&want[j], &got[j], 4) != 0) { std::fprintf(stderr, "buffer %d (%d frames): instance %zu differs first at float %zu\n", k, frames, j / (static_cast<size_t>(frames) * ch), j); break; }
            return 1;
        }
        ++k;
    }
    // errors name the device and the range
    if (oalsfx_group_set_effect_type(g, 0, n + 1, 0, 1) || !std::strstr(oalsfx_group_error(g), "range")) { std::fprintf(stderr, "range check: %s\n", oalsfx_group_error(g)); return 1; }
    oalsfx_group_destroy(g);
    oalsfx_batch_destroy(b);
    std::printf("ok\n");
    return 0;
}
