// oalsfxpp::Api -- the drop-in C++ facade (include/oalsfxpp.h), implemented as a batch of one
// instance on the current HIP device.  Argument checks, return values and error strings follow the
// reference facade (reference src/oalsfxpp.cpp:3449-3903); the sample path is the HIP backend.
#include <cstdlib>
#include <new>

#include "core.hpp"
#include "oalsfx_hip.h"

namespace oalsfxpp {

namespace {

// reference ApiErrorMessages (src/oalsfxpp.cpp:3449-3457)
constexpr const char* err_none = "";
constexpr const char* err_allocate = "Failed to allocate implementaion class.";
constexpr const char* err_not_initialized = "Not initialized.";
constexpr const char* err_index = "Effect index is out of range.";
constexpr const char* err_no_src = "No source samples.";
constexpr const char* err_no_dst = "No destination samples.";

} // namespace

class Api::Impl {
public:
    oalsfx_batch* batch = nullptr;
    ChannelFormat format = ChannelFormat::none;
    int rate = 0;
    int channels = 0;
    int effect_count = 0;
    const char* error_message = err_none;

    ~Impl()
    {
        if (batch) oalsfx_batch_destroy(batch);
    }
};

Api::Api() : pimpl_{}, error_message_{err_none} {}

Api::~Api() { uninitialize(); }

bool Api::initialize(const ChannelFormat channel_format, const int sampling_rate, const int effect_count)
{
    uninitialize();
    pimpl_.reset(new (std::nothrow) Impl{});
    if (!pimpl_) {
        error_message_ = err_allocate;
        return false;
    }
    // The reference's initialize has no notion of a device.  A relinked caller places its instances with OALSFX_DEVICE=<HIP
    // ordinal> (read at every initialize, so a process may move between calls); default 0.  A value that is not a
    // valid ordinal fails in oalsfx_batch_create with its message.
    int device = 0;
    if (const char* env = std::getenv("OALSFX_DEVICE")) {
        char* end = nullptr;
        const long v = std::strtol(env, &end, 10);
        device = (end != env && *end == 0 && v >= 0 && v < (1 << 16)) ? static_cast<int>(v) : -1;
    }
    pimpl_->batch = oalsfx_batch_create(1, static_cast<int>(channel_format), sampling_rate, effect_count, device);
    if (!pimpl_->batch) {
        // the message of a failed create lives in thread-local storage of the library; keep a static
        // string for the three argument errors the reference distinguishes (src/oalsfxpp.cpp:2808-2810)
        const char* msg = oalsfx_last_error();
        static thread_local char copy[160];
        int i = 0;
        for (; msg && msg[i] && i < 159; ++i) copy[i] = msg[i];
        copy[i] = 0;
        error_message_ = copy;
        uninitialize();
        return false;
    }
    pimpl_->format = channel_format;
    pimpl_->rate = sampling_rate;
    pimpl_->channels = oalsfx_batch_channels(pimpl_->batch);
    pimpl_->effect_count = effect_count;
    return true;
}

bool Api::is_initialized() const { return pimpl_ != nullptr; }

#define REQUIRE_INIT(ret)                       \
    if (!is_initialized()) {                    \
        error_message_ = err_not_initialized;   \
        return ret;                             \
    }

int Api::get_sampling_rate() const
{
    REQUIRE_INIT(0)
    return pimpl_->rate;
}

ChannelFormat Api::get_channel_format() const
{
    REQUIRE_INIT(ChannelFormat::none)
    return pimpl_->format;
}

int Api::get_channel_count() const
{
    REQUIRE_INIT(0)
    return pimpl_->channels;
}

int Api::get_effect_count() const
{
    REQUIRE_INIT(0)
    return pimpl_->effect_count;
}

bool Api::get_effect(const int effect_index, Effect& effect) const
{
    REQUIRE_INIT(false)
    if (effect_index < 0 || effect_index >= pimpl_->effect_count) {
        error_message_ = err_index;
        return false;
    }
    return oalsfx_batch_get_effect(pimpl_->batch, 0, effect_index, 0, reinterpret_cast<oalsfx_effect*>(&effect)) != 0;
}

bool Api::get_deferred_effect(const int effect_index, Effect& effect) const
{
    REQUIRE_INIT(false)
    if (effect_index < 0 || effect_index >= pimpl_->effect_count) {
        error_message_ = err_index;
        return false;
    }
    return oalsfx_batch_get_effect(pimpl_->batch, 0, effect_index, 1, reinterpret_cast<oalsfx_effect*>(&effect)) != 0;
}

bool Api::set_effect_type(const int effect_index, const EffectType effect_type)
{
    REQUIRE_INIT(false)
    if (effect_index < 0 || effect_index >= pimpl_->effect_count) {
        error_message_ = err_index;
        return false;
    }
    return oalsfx_batch_set_effect_type(pimpl_->batch, 0, 1, effect_index, static_cast<int>(effect_type)) != 0;
}

bool Api::set_effect_props(const int effect_index, const EffectProps& effect_props)
{
    REQUIRE_INIT(false)
    if (effect_index < 0 || effect_index >= pimpl_->effect_count) {
        error_message_ = err_index;
        return false;
    }
    return oalsfx_batch_set_effect_props(pimpl_->batch, 0, 1, effect_index, &effect_props, 0) != 0;
}

bool Api::set_effect(const int effect_index, const Effect& effect)
{
    REQUIRE_INIT(false)
    if (effect_index < 0 || effect_index >= pimpl_->effect_count) {
        error_message_ = err_index;
        return false;
    }
    oalsfx_batch_set_effect(pimpl_->batch, 0, 1, effect_index, reinterpret_cast<const oalsfx_effect*>(&effect), 0);
    // The reference stores the effect and then reports failure (src/oalsfxpp.cpp:3655-3657); callers that
    // ignore the result, like the reference's own test program, depend on the store only.  Kept as is.
    return false;
}

bool Api::get_send_props(const int effect_index, SendProps& send_props) const
{
    REQUIRE_INIT(false)
    if (effect_index >= pimpl_->effect_count) {
        error_message_ = err_index;
        return false;
    }
    return oalsfx_batch_get_send_props(pimpl_->batch, 0, effect_index, 0, reinterpret_cast<oalsfx_send_props*>(&send_props)) != 0;
}

bool Api::get_deferred_send_props(const int effect_index, SendProps& send_props) const
{
    REQUIRE_INIT(false)
    if (effect_index >= pimpl_->effect_count) {
        error_message_ = err_index;
        return false;
    }
    return oalsfx_batch_get_send_props(pimpl_->batch, 0, effect_index, 1, reinterpret_cast<oalsfx_send_props*>(&send_props)) != 0;
}

bool Api::set_send_props(const int effect_index, const SendProps& send_props)
{
    REQUIRE_INIT(false)
    if (effect_index >= pimpl_->effect_count) {
        error_message_ = err_index;
        return false;
    }
    return oalsfx_batch_set_send_props(pimpl_->batch, 0, 1, effect_index, reinterpret_cast<const oalsfx_send_props*>(&send_props)) != 0;
}

bool Api::apply_changes()
{
    REQUIRE_INIT(false)
    return oalsfx_batch_apply_changes(pimpl_->batch, 0, 1) != 0;
}

bool Api::mix(const int sample_count, const float* src_samples, float* dst_samples)
{
    REQUIRE_INIT(false)
    if (sample_count == 0) return true;
    if (!src_samples) {
        error_message_ = err_no_src;
        return false;
    }
    if (!dst_samples) {
        error_message_ = err_no_dst;
        return false;
    }
    if (!oalsfx_batch_mix(pimpl_->batch, sample_count, src_samples, dst_samples)) {
        error_message_ = oalsfx_batch_error(pimpl_->batch);
        return false;
    }
    return true;
}

void Api::uninitialize() { pimpl_ = nullptr; }

const char* Api::get_error_message() const
{
    // The reference reads the message through pimpl_ and crashes when the instance was never initialised
    // (src/oalsfxpp.cpp:3836-3839); this facade returns the facade-level message instead of crashing.
    return error_message_;
}

int Api::get_min_channels() { return 1; }
int Api::get_max_channels() { return OALSFX_MAX_CHANNELS; }
int Api::get_min_sampling_rate() { return oalsfx_host::min_sampling_rate; }
int Api::get_max_sampling_rate() { return oalsfx_host::max_sampling_rate; }
int Api::get_min_effects() { return 1; }
int Api::get_max_effects() { return OALSFX_MAX_SLOTS; }

ChannelFormat Api::channel_count_to_channel_format(const int channel_count)
{
    switch (channel_count) {
    case 1: return ChannelFormat::mono;
    case 2: return ChannelFormat::stereo;
    case 4: return ChannelFormat::quad;
    case 6: return ChannelFormat::five_point_one;
    case 7: return ChannelFormat::six_point_one;
    case 8: return ChannelFormat::seven_point_one;
    default: return ChannelFormat::none;
    }
}

int Api::channel_format_to_channel_count(const ChannelFormat channel_format) { return oalsfx_host::channel_count_of(channel_format); }

} // namespace oalsfxpp
