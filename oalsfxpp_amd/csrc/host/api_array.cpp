// oalsfxpp::ApiArray (include/oalsfxpp_array.h): many Api objects' worth of effect chains as one batch.  Argument checks, return
// values and messages follow Api's (api.cpp; reference src/oalsfxpp.cpp:3449-3903).
#include <cstdlib>

#include "oalsfx_hip.h"
#include "oalsfxpp_array.h"

namespace oalsfxpp {

namespace {
constexpr const char* err_none = "";
constexpr const char* err_not_initialized = "Not initialized.";
constexpr const char* err_index = "Effect index is out of range.";
constexpr const char* err_instance = "Instance index is out of range.";
constexpr const char* err_no_src = "No source samples.";
constexpr const char* err_no_dst = "No destination samples.";

const oalsfx_effect* as_c(const Effect& e) { return reinterpret_cast<const oalsfx_effect*>(&e); }
static_assert(sizeof(Effect) == sizeof(oalsfx_effect), "oalsfx_effect mirrors oalsfxpp::Effect");
static_assert(sizeof(SendProps) == sizeof(oalsfx_send_props), "oalsfx_send_props mirrors oalsfxpp::SendProps");
} // namespace

ApiArray::ApiArray() : batch_(nullptr), count_(0), channels_(0), effects_(0), error_(err_none) {}
ApiArray::~ApiArray() { uninitialize(); }

bool ApiArray::initialize(int count, ChannelFormat channel_format, int sampling_rate, int effect_count, int device)
{
    uninitialize();
    if (device < 0) {
        device = 0;
        if (const char* env = std::getenv("OALSFX_DEVICE")) {
            char* end = nullptr;
            const long v = std::strtol(env, &end, 10);
            device = (end != env && *end == 0 && v >= 0 && v < (1 << 16)) ? static_cast<int>(v) : -1;
        }
    }
    batch_ = oalsfx_batch_create(count, static_cast<int>(channel_format), sampling_rate, effect_count, device);
    if (!batch_) {
        error_ = oalsfx_last_error();
        return false;
    }
    count_ = count;
    channels_ = oalsfx_batch_channels(batch_);
    effects_ = effect_count;
    error_ = err_none;
    return true;
}

bool ApiArray::is_initialized() const { return batch_ != nullptr; }

void ApiArray::uninitialize()
{
    if (batch_) oalsfx_batch_destroy(batch_);
    batch_ = nullptr;
    count_ = channels_ = effects_ = 0;
}

int ApiArray::size() const { return count_; }
int ApiArray::get_channel_count() const { return channels_; }
int ApiArray::get_effect_count() const { return effects_; }
const char* ApiArray::get_error_message() const { return error_; }
oalsfx_batch* ApiArray::batch() const { return batch_; }

#define OALSFXPP_ARRAY_CHECK(index, effect_index, allow_direct)                                        \
    if (!batch_) { error_ = err_not_initialized; return false; }                                       \
    if ((index) < 0 || (index) >= count_) { error_ = err_instance; return false; }                      \
    if ((effect_index) >= effects_ || (!(allow_direct) && (effect_index) < 0)) { error_ = err_index; return false; }

bool ApiArray::get_effect(int index, int effect_index, Effect& effect) const
{
    OALSFXPP_ARRAY_CHECK(index, effect_index, false);
    return oalsfx_batch_get_effect(batch_, index, effect_index, 0, reinterpret_cast<oalsfx_effect*>(&effect)) != 0;
}

bool ApiArray::get_deferred_effect(int index, int effect_index, Effect& effect) const
{
    OALSFXPP_ARRAY_CHECK(index, effect_index, false);
    return oalsfx_batch_get_effect(batch_, index, effect_index, 1, reinterpret_cast<oalsfx_effect*>(&effect)) != 0;
}

bool ApiArray::set_effect_type(int index, int effect_index, EffectType effect_type)
{
    OALSFXPP_ARRAY_CHECK(index, effect_index, false);
    if (!oalsfx_batch_set_effect_type(batch_, index, 1, effect_index, static_cast<int>(effect_type))) { error_ = oalsfx_batch_error(batch_); return false; }
    return true;
}

bool ApiArray::set_effect_props(int index, int effect_index, const EffectProps& effect_props)
{
    OALSFXPP_ARRAY_CHECK(index, effect_index, false);
    if (!oalsfx_batch_set_effect_props(batch_, index, 1, effect_index, &effect_props, 0)) { error_ = oalsfx_batch_error(batch_); return false; }
    return true;
}

bool ApiArray::set_effect(int index, int effect_index, const Effect& effect)
{
    OALSFXPP_ARRAY_CHECK(index, effect_index, false);
    if (!oalsfx_batch_set_effect(batch_, index, 1, effect_index, as_c(effect), 0)) error_ = oalsfx_batch_error(batch_);
    return false; // (the reference's Api::set_effect returns false on success too, src/oalsfxpp.cpp:3657)
}

bool ApiArray::set_send_props(int index, int effect_index, const SendProps& send_props)
{
    OALSFXPP_ARRAY_CHECK(index, effect_index, true);
    if (!oalsfx_batch_set_send_props(batch_, index, 1, effect_index, reinterpret_cast<const oalsfx_send_props*>(&send_props))) { error_ = oalsfx_batch_error(batch_); return false; }
    return true;
}

bool ApiArray::set_effect_type_all(int effect_index, EffectType effect_type)
{
    OALSFXPP_ARRAY_CHECK(0, effect_index, false);
    if (!oalsfx_batch_set_effect_type(batch_, 0, count_, effect_index, static_cast<int>(effect_type))) { error_ = oalsfx_batch_error(batch_); return false; }
    return true;
}

bool ApiArray::set_effect_all(int effect_index, const Effect& effect)
{
    OALSFXPP_ARRAY_CHECK(0, effect_index, false);
    if (!oalsfx_batch_set_effect(batch_, 0, count_, effect_index, as_c(effect), 0)) error_ = oalsfx_batch_error(batch_);
    return false;
}

bool ApiArray::apply_changes()
{
    if (!batch_) { error_ = err_not_initialized; return false; }
    if (!oalsfx_batch_apply_changes(batch_, 0, count_)) { error_ = oalsfx_batch_error(batch_); return false; }
    return true;
}

bool ApiArray::apply_changes(int index)
{
    OALSFXPP_ARRAY_CHECK(index, 0, false);
    if (!oalsfx_batch_apply_changes(batch_, index, 1)) { error_ = oalsfx_batch_error(batch_); return false; }
    return true;
}

bool ApiArray::mix(int sample_count, const float* const* src_samples, float* const* dst_samples)
{
    // Api::mix's preconditions (reference src/oalsfxpp.cpp:3790-3811)
    if (!batch_) { error_ = err_not_initialized; return false; }
    if (sample_count == 0) return true;
    if (!src_samples) { error_ = err_no_src; return false; }
    if (!dst_samples) { error_ = err_no_dst; return false; }
    if (!oalsfx_batch_mix_gather(batch_, sample_count, src_samples, dst_samples)) { error_ = oalsfx_batch_error(batch_); return false; }
    return true;
}

bool ApiArray::mix(int sample_count, const float* src_samples, float* dst_samples)
{
    if (!batch_) { error_ = err_not_initialized; return false; }
    if (sample_count == 0) return true;
    if (!src_samples) { error_ = err_no_src; return false; }
    if (!dst_samples) { error_ = err_no_dst; return false; }
    if (!oalsfx_batch_mix(batch_, sample_count, src_samples, dst_samples)) { error_ = oalsfx_batch_error(batch_); return false; }
    return true;
}

} // namespace oalsfxpp
