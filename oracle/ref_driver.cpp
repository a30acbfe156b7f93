// TEST INFRASTRUCTURE -- not part of the product.
//
// C-ABI shim around the *untouched* reference implementation.  The reference
// translation unit is compiled from where it lies (/root/reference/src, passed
// by the Makefile as REF_SRC) and is never copied into this repository.  The
// shim only exists in this container: the output (oracle/_ref/libref.so) is
// git-ignored and nothing under tests -m gpu / smoke / bench reads
// /root/reference at run time.
//
// Besides the public oalsfxpp::Api calls, the shim dumps the reference's private
// derived state into the repository's descriptor structs (include/oalsfx_desc.h)
// so that the host update path and the process path can be pinned separately.

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <thread>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <new>
#include <type_traits>
#include <vector>

// Open up the reference's private members for the dumps (std headers above are
// already included, so only the reference's own classes are affected).
#define private public
#define protected public
#include REF_SRC
#undef private
#undef protected

#include "oalsfx_desc.h"

using namespace oalsfxpp;

namespace {

void copy_coef(const FilterState& f, oalsfx_biquad_t& o)
{
    o.b0 = f.b0_; o.b1 = f.b1_; o.b2 = f.b2_; o.a1 = f.a1_; o.a2 = f.a2_;
}

void copy_hist(const FilterState& f, oalsfx_hist_t& o)
{
    o.x[0] = f.x_[0]; o.x[1] = f.x_[1]; o.y[0] = f.y_[0]; o.y[1] = f.y_[1];
}

template <typename G>
void copy_gains(const G& g, float* o)
{
    for (int i = 0; i < OALSFX_MAX_CHANNELS; ++i) o[i] = g[i];
}

void dump_send(const Source::Send& s, int in_channels, oalsfx_send_params& o)
{
    std::memset(&o, 0, sizeof(o));
    o.filter_type = static_cast<int>(s.filter_type_);
    o.out_channels = s.buffers_ ? s.channel_count_ : 0;
    copy_coef(s.channels_[0].low_pass_, o.lp);
    copy_coef(s.channels_[0].high_pass_, o.hp);
    for (int c = 0; c < in_channels; ++c) copy_gains(s.channels_[c].target_gains_, o.gains[c]);
}

struct Preset { const char* name; const EffectProps::Reverb* props; };

#define OALSFX_PRESET(group, name) {#group "::" #name, &ReverbPresets::group::name},
const Preset g_presets[] = {
#include "oalsfx_preset_names.inc"
};
#undef OALSFX_PRESET

} // namespace

extern "C" {

int ref_sizeof_effect() { return static_cast<int>(sizeof(Effect)); }
int ref_sizeof_props() { return static_cast<int>(sizeof(EffectProps)); }

void* ref_create(int channel_format, int rate, int effect_count)
{
    auto api = new Api{};
    if (!api->initialize(static_cast<ChannelFormat>(channel_format), rate, effect_count)) {
        delete api;
        return nullptr;
    }
    return api;
}

void ref_destroy(void* h) { delete static_cast<Api*>(h); }

int ref_set_effect(void* h, int idx, const void* effect)
{
    Effect e;
    std::memcpy(&e, effect, sizeof(Effect));
    return static_cast<Api*>(h)->set_effect(idx, e) ? 1 : 0;
}

int ref_set_effect_type(void* h, int idx, int type)
{
    return static_cast<Api*>(h)->set_effect_type(idx, static_cast<EffectType>(type)) ? 1 : 0;
}

int ref_set_effect_props(void* h, int idx, const void* props)
{
    EffectProps p;
    std::memcpy(&p, props, sizeof(EffectProps));
    return static_cast<Api*>(h)->set_effect_props(idx, p) ? 1 : 0;
}

int ref_get_effect(void* h, int idx, int deferred, void* effect)
{
    Effect e;
    std::memset(&e, 0, sizeof(e));
    const bool ok = deferred ? static_cast<Api*>(h)->get_deferred_effect(idx, e) : static_cast<Api*>(h)->get_effect(idx, e);
    std::memcpy(effect, &e, sizeof(Effect));
    return ok ? 1 : 0;
}

int ref_set_send_props(void* h, int idx, const float* p)
{
    SendProps s;
    s.gain_ = p[0]; s.gain_hf_ = p[1]; s.gain_lf_ = p[2];
    return static_cast<Api*>(h)->set_send_props(idx, s) ? 1 : 0;
}

int ref_get_send_props(void* h, int idx, int deferred, float* p)
{
    SendProps s{};
    const bool ok = deferred ? static_cast<Api*>(h)->get_deferred_send_props(idx, s) : static_cast<Api*>(h)->get_send_props(idx, s);
    p[0] = s.gain_; p[1] = s.gain_hf_; p[2] = s.gain_lf_;
    return ok ? 1 : 0;
}

int ref_apply_changes(void* h) { return static_cast<Api*>(h)->apply_changes() ? 1 : 0; }

int ref_mix(void* h, int frames, const float* src, float* dst)
{
    return static_cast<Api*>(h)->mix(frames, src, dst) ? 1 : 0;
}

const char* ref_error(void* h) { return static_cast<Api*>(h)->get_error_message(); }

// Effect::set_type_and_defaults / Effect::normalize on a caller-owned Effect.
void ref_effect_defaults(int type, void* effect)
{
    Effect e;
    std::memset(&e, 0, sizeof(e));
    e.set_type_and_defaults(static_cast<EffectType>(type));
    std::memcpy(effect, &e, sizeof(Effect));
}

void ref_effect_normalize(void* effect)
{
    Effect e;
    std::memcpy(&e, effect, sizeof(Effect));
    e.normalize();
    std::memcpy(effect, &e, sizeof(Effect));
}

int ref_preset_count() { return static_cast<int>(sizeof(g_presets) / sizeof(g_presets[0])); }
const char* ref_preset_name(int i) { return g_presets[i].name; }
void ref_preset_props(int i, void* reverb_props) { std::memcpy(reverb_props, g_presets[i].props, sizeof(EffectProps::Reverb)); }
int ref_sizeof_reverb_props() { return static_cast<int>(sizeof(EffectProps::Reverb)); }

// Runs the lazy parameter refresh that mix_data would run first
// (update_context_sources, reference src/oalsfxpp.cpp:3397) so the dumps below see
// the derived state of the next mix call.  Idempotent with respect to the output.
void ref_refresh(void* h) { static_cast<Api*>(h)->pimpl_->update_context_sources(); }

int ref_channel_count(void* h) { return static_cast<Api*>(h)->pimpl_->device_.channel_count_; }

void ref_dump_source(void* h, oalsfx_source_params* p, oalsfx_source_state* s)
{
    auto& impl = *static_cast<Api*>(h)->pimpl_;
    const int ch = impl.device_.channel_count_;
    std::memset(p, 0, sizeof(*p));
    std::memset(s, 0, sizeof(*s));
    dump_send(impl.source_.direct_, ch, p->direct);
    for (int c = 0; c < ch; ++c) {
        copy_hist(impl.source_.direct_.channels_[c].low_pass_, s->lp[0][c]);
        copy_hist(impl.source_.direct_.channels_[c].high_pass_, s->hp[0][c]);
    }
    for (int i = 0; i < impl.effect_count_; ++i) {
        dump_send(impl.source_.auxes_[i], ch, p->aux[i]);
        for (int c = 0; c < ch; ++c) {
            copy_hist(impl.source_.auxes_[i].channels_[c].low_pass_, s->lp[1 + i][c]);
            copy_hist(impl.source_.auxes_[i].channels_[c].high_pass_, s->hp[1 + i][c]);
        }
    }
}

// Dumps slot `idx`.  Ring layout fields (ring_len / ring_off) are filled with the
// repository's convention (oalsfx_reverb_place_rings): longest ring first, 4 contiguous lines each.
int ref_dump_slot(void* h, int idx, oalsfx_slot_params* p, oalsfx_slot_state* s)
{
    auto& impl = *static_cast<Api*>(h)->pimpl_;
    auto& slot = impl.effect_contexts_[idx].effect_slot_;
    EffectState* st = slot.effect_state_.get();
    std::memset(p, 0, sizeof(*p));
    std::memset(s, 0, sizeof(*s));
    p->type = static_cast<int>(slot.effect_.type_);

    switch (slot.effect_.type_) {
    case EffectType::null:
        break;
    case EffectType::chorus: {
        auto& e = *static_cast<ChorusEffectState*>(st);
        auto& o = p->u.moddelay;
        o.waveform = static_cast<int>(e.waveform_); o.delay = e.delay_; o.depth = e.depth_; o.feedback = e.feedback_;
        o.lfo_range = e.lfo_range_; o.lfo_scale = e.lfo_scale_; o.lfo_disp = e.lfo_disp_; o.ring_len = e.buffer_length_;
        copy_gains(e.sides_gains_[0], o.gains[0]); copy_gains(e.sides_gains_[1], o.gains[1]);
        s->u.moddelay.offset = e.offset_;
        break;
    }
    case EffectType::flanger: {
        auto& e = *static_cast<FlangerEffectState*>(st);
        auto& o = p->u.moddelay;
        o.waveform = static_cast<int>(e.waveform_); o.delay = e.delay_; o.depth = e.depth_; o.feedback = e.feedback_;
        o.lfo_range = e.lfo_range_; o.lfo_scale = e.lfo_scale_; o.lfo_disp = e.lfo_disp_; o.ring_len = e.buffer_length_;
        copy_gains(e.sides_gains_[0], o.gains[0]); copy_gains(e.sides_gains_[1], o.gains[1]);
        s->u.moddelay.offset = e.offset_;
        break;
    }
    case EffectType::compressor: {
        auto& e = *static_cast<CompressorEffectState*>(st);
        auto& o = p->u.compressor;
        o.enabled = e.is_enabled_ ? 1 : 0; o.attack_rate = e.attack_rate_; o.release_rate = e.release_rate_;
        for (int i = 0; i < 4; ++i) copy_gains(e.channels_gains_[i], o.gains[i]);
        s->u.compressor.gain_control = e.gain_control_;
        break;
    }
    case EffectType::dedicated_dialog:
    case EffectType::dedicated_low_frequency: {
        auto& e = *static_cast<DedicatedEffectState*>(st);
        copy_gains(e.gains_, p->u.dedicated.gains);
        break;
    }
    case EffectType::distortion: {
        auto& e = *static_cast<DistortionEffectState*>(st);
        auto& o = p->u.distortion;
        copy_coef(e.low_pass_, o.low_pass); copy_coef(e.band_pass_, o.band_pass);
        o.attenuation = e.attenuation_; o.edge_coeff = e.edge_coeff_;
        copy_gains(e.gains_, o.gains);
        copy_hist(e.low_pass_, s->u.distortion.low_pass); copy_hist(e.band_pass_, s->u.distortion.band_pass);
        break;
    }
    case EffectType::echo: {
        auto& e = *static_cast<EchoEffectState*>(st);
        auto& o = p->u.echo;
        o.tap1 = e.taps_[0].delay; o.tap2 = e.taps_[1].delay; o.feed_gain = e.feed_gain_; o.ring_len = e.buffer_length_;
        copy_coef(e.filter_, o.filter);
        copy_gains(e.taps_gains_[0], o.gains[0]); copy_gains(e.taps_gains_[1], o.gains[1]);
        s->u.echo.offset = e.offset_; copy_hist(e.filter_, s->u.echo.filter);
        break;
    }
    case EffectType::equalizer: {
        auto& e = *static_cast<EqualizerEffectState*>(st);
        auto& o = p->u.equalizer;
        for (int b = 0; b < 4; ++b) {
            copy_coef(e.filter_[b][0], o.band[b]);
            for (int c = 0; c < 4; ++c) copy_hist(e.filter_[b][c], s->u.equalizer.hist[b][c]);
        }
        for (int i = 0; i < 4; ++i) copy_gains(e.channels_gains_[i], o.gains[i]);
        break;
    }
    case EffectType::ring_modulator: {
        auto& e = *static_cast<RingModulatorEffectState*>(st);
        auto& o = p->u.ringmod;
        o.waveform = (e.process_func_ == RingModulatorEffectState::modulate_sin) ? 0 :
                     (e.process_func_ == RingModulatorEffectState::modulate_saw) ? 1 : 2;
        o.step = e.step_;
        copy_coef(e.filters_[0], o.filter);
        for (int i = 0; i < 4; ++i) { copy_gains(e.channels_gains_[i], o.gains[i]); copy_hist(e.filters_[i], s->u.ringmod.hist[i]); }
        s->u.ringmod.index = e.index_;
        break;
    }
    case EffectType::reverb:
    case EffectType::eax_reverb: {
        auto& e = *static_cast<ReverbEffectState*>(st);
        auto& o = p->u.reverb;
        auto& q = s->u.reverb;
        o.is_eax = e.is_eax_ ? 1 : 0;
        copy_coef(e.filters_[0].lp_, o.lp); copy_coef(e.filters_[0].hp_, o.hp);
        o.late_feed_tap = e.late_feed_tap_;
        o.ap_feed_coeff = e.ap_feed_coeff_; o.mix_x = e.mix_x_; o.mix_y = e.mix_y_;
        o.mod_range = e.mod_.range_; o.mod_depth = e.mod_.depth_; o.mod_coeff = e.mod_.coeff_;
        o.density_gain = e.late_.density_gain_;
        for (int j = 0; j < 4; ++j) {
            o.early_tap[j] = e.early_delay_taps_[j][1]; q.cur_early_tap[j] = e.early_delay_taps_[j][0];
            o.early_tap_coeff[j] = e.early_delay_coeffs_[j];
            o.late_tap[j] = e.late_delay_taps_[j][1]; q.cur_late_tap[j] = e.late_delay_taps_[j][0];
            o.early_ap_off[j] = e.early_.vec_ap_.offsets_[j][1]; q.cur_early_ap_off[j] = e.early_.vec_ap_.offsets_[j][0];
            o.early_line_off[j] = e.early_.offsets_[j][1]; q.cur_early_line_off[j] = e.early_.offsets_[j][0];
            o.early_line_coeff[j] = e.early_.coeffs_[j];
            o.late_line_off[j] = e.late_.offsets_[j][1]; q.cur_late_line_off[j] = e.late_.offsets_[j][0];
            o.late_ap_off[j] = e.late_.vec_ap_.offsets_[j][1]; q.cur_late_ap_off[j] = e.late_.vec_ap_.offsets_[j][0];
            for (int k = 0; k < 3; ++k) { o.t60_lf[j][k] = e.late_.filters_[j].lf_coeffs_[k]; o.t60_hf[j][k] = e.late_.filters_[j].hf_coeffs_[k]; }
            o.t60_mid[j] = e.late_.filters_[j].mid_coeff_;
            for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) q.t60[j][a][b] = e.late_.filters_[j].states_[a][b];
            copy_gains(e.early_.pan_gains_[j], o.early_pan[j]); copy_gains(e.late_.pan_gains_[j], o.late_pan[j]);
            copy_gains(e.early_.current_gains_[j], q.early_cur_gain[j]); copy_gains(e.late_.current_gains_[j], q.late_cur_gain[j]);
            copy_hist(e.filters_[j].lp_, q.lp[j]); copy_hist(e.filters_[j].hp_, q.hp[j]);
        }
        const ReverbEffectState::DelayLineI* rings[5] = {
            &e.delay_, &e.early_.vec_ap_.delay_, &e.early_.delay_, &e.late_.vec_ap_.delay_, &e.late_.delay_};
        for (int r = 0; r < 5; ++r) o.ring_len[r] = rings[r]->get_sample_count();
        oalsfx_reverb_place_rings(o.ring_len, o.ring_off);
        q.mod_index = e.mod_.index_; q.mod_range = e.mod_.range_; q.mod_filter = e.mod_.filter_;
        q.fade_count = e.fade_count_; q.offset = e.offset_;
        break;
    }
    }
    return 1;
}

// Copies one delay ring of slot `idx` into `out` using the repository's layout
// (line-major: for reverb rings 4 lines of ring_len floats; chorus/flanger 2 sides;
// echo 1 line).  Returns the number of floats written (0 if the slot has no ring r).
int ref_dump_ring(void* h, int idx, int r, float* out)
{
    auto& impl = *static_cast<Api*>(h)->pimpl_;
    auto& slot = impl.effect_contexts_[idx].effect_slot_;
    EffectState* st = slot.effect_state_.get();
    switch (slot.effect_.type_) {
    case EffectType::chorus: {
        auto& e = *static_cast<ChorusEffectState*>(st);
        if (r != 0) return 0;
        for (int side = 0; side < 2; ++side) std::copy(e.sample_buffers_[side].begin(), e.sample_buffers_[side].end(), out + side * e.buffer_length_);
        return 2 * e.buffer_length_;
    }
    case EffectType::flanger: {
        auto& e = *static_cast<FlangerEffectState*>(st);
        if (r != 0) return 0;
        for (int side = 0; side < 2; ++side) std::copy(e.sample_buffers_[side].begin(), e.sample_buffers_[side].end(), out + side * e.buffer_length_);
        return 2 * e.buffer_length_;
    }
    case EffectType::echo: {
        auto& e = *static_cast<EchoEffectState*>(st);
        if (r != 0) return 0;
        std::copy(e.sample_buffer_.begin(), e.sample_buffer_.end(), out);
        return e.buffer_length_;
    }
    case EffectType::reverb:
    case EffectType::eax_reverb: {
        auto& e = *static_cast<ReverbEffectState*>(st);
        const ReverbEffectState::DelayLineI* rings[5] = {
            &e.delay_, &e.early_.vec_ap_.delay_, &e.early_.delay_, &e.late_.vec_ap_.delay_, &e.late_.delay_};
        if (r < 0 || r > 4) return 0;
        const int n = rings[r]->get_sample_count();
        for (int j = 0; j < 4; ++j) for (int i = 0; i < n; ++i) out[j * n + i] = rings[r]->lines_[i][j];
        return 4 * n;
    }
    default:
        return 0;
    }
}

// cpu_baseline (kind "reference"): n_instances reference Api objects, each with `effect` in slot 0, fed the synthetic
// input of SURVEY 8d (same generator as oracle_synth / k_fill_synthetic) for warmup + buffers calls of Api::mix on
// `threads` host threads over a dynamic partition of the instances.  Returns the seconds the timed buffers took.
double ref_bench(int channel_format, int rate, const void* effect, int n_instances, int frames, int warmup, int buffers, int threads)
{
    std::vector<std::unique_ptr<Api>> inst(n_instances);
    int channels = 0;
    for (auto& p : inst) {
        p.reset(new Api{});
        if (!p->initialize(static_cast<ChannelFormat>(channel_format), rate, 1)) return -1.0;
        Effect e;
        std::memcpy(&e, effect, sizeof(Effect));
        p->set_effect(0, e);
        p->apply_changes();
        channels = p->get_channel_count();
    }
    auto synth = [](uint32_t instance, uint32_t buffer_index, float* out, int count) {
        uint32_t x = 0x9E3779B9u ^ (instance * 2654435761u) ^ buffer_index;
        if (x == 0) x = 1;
        for (int i = 0; i < count; ++i) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            out[i] = static_cast<float>(x >> 8) * (1.0F / 8388608.0F) - 1.0F;
        }
    };
    auto run = [&](int first_buf, int count) {
        std::atomic<int> next{0};
        auto worker = [&]() {
            std::vector<float> src(static_cast<size_t>(frames) * channels), dst(src.size());
            for (;;) {
                const int i = next.fetch_add(1);
                if (i >= n_instances) break;
                for (int b = 0; b < count; ++b) {
                    synth(static_cast<uint32_t>(i), static_cast<uint32_t>(first_buf + b), src.data(), static_cast<int>(src.size()));
                    inst[i]->mix(frames, src.data(), dst.data());
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
    return std::chrono::duration<double>(t1 - t0).count();
}

} // extern "C"
