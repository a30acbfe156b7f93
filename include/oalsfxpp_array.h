// oalsfxpp::ApiArray -- many Api objects' worth of effect chains advanced together.
//
// The reference has one class, oalsfxpp::Api (src/oalsfxpp.h:760-922): one effect chain, one mix call per buffer.  A program that
// holds thousands of them -- a voice per object -- and relinks against this library gets thousands of one-instance GPU batches, one
// wavefront per launch.  ApiArray is the same surface for `count` chains at once: the setters take the instance's index in front of
// the reference's arguments (same meaning, same return values, same messages), apply_changes and mix act on all of them, and mix takes
// either one interleaved buffer pair for the whole array or one pair per instance (what the per-object code already has).  Everything
// goes through the batch C ABI (include/oalsfx_hip.h); nothing here is needed by code that keeps using Api.
#ifndef OALSFXPP_ARRAY_H
#define OALSFXPP_ARRAY_H

#include "oalsfxpp.h"

struct oalsfx_batch;

namespace oalsfxpp {

class ApiArray {
public:
    ApiArray();
    ApiArray(const ApiArray&) = delete;
    ApiArray& operator=(const ApiArray&) = delete;
    ~ApiArray();

    // `count` chains of one format, rate and effect count (Api::initialize's checks and messages); device: HIP ordinal
    // (-1: OALSFX_DEVICE or 0, as Api does).
    bool initialize(int count, ChannelFormat channel_format, int sampling_rate, int effect_count, int device = -1);
    bool is_initialized() const;
    void uninitialize();
    int size() const;
    int get_channel_count() const;
    int get_effect_count() const;
    const char* get_error_message() const;

    // Api's deferred setters and getters, for instance `index`.  set_effect returns false on success like the reference's
    // (src/oalsfxpp.cpp:3657); set_send_props with effect_index < 0 addresses the direct send.
    bool get_effect(int index, int effect_index, Effect& effect) const;
    bool get_deferred_effect(int index, int effect_index, Effect& effect) const;
    bool set_effect_type(int index, int effect_index, EffectType effect_type);
    bool set_effect_props(int index, int effect_index, const EffectProps& effect_props);
    bool set_effect(int index, int effect_index, const Effect& effect);
    bool set_send_props(int index, int effect_index, const SendProps& send_props);
    // ... and the same value for every instance in one call
    bool set_effect_type_all(int effect_index, EffectType effect_type);
    bool set_effect_all(int effect_index, const Effect& effect);

    bool apply_changes();          // Api::apply_changes on every instance
    bool apply_changes(int index); // ... on one

    // Api::mix for every instance: one buffer pair per instance (sample_count * channels floats each) ...
    bool mix(int sample_count, const float* const* src_samples, float* const* dst_samples);
    // ... or the whole array in one interleaved pair, [instance][frame][channel]
    bool mix(int sample_count, const float* src_samples, float* dst_samples);

    oalsfx_batch* batch() const; // for what the C ABI offers beyond this (device-resident buffers, pipelined host calls, read-backs)

private:
    oalsfx_batch* batch_;
    int count_, channels_, effects_;
    mutable const char* error_;
};

} // namespace oalsfxpp

#endif // OALSFXPP_ARRAY_H
