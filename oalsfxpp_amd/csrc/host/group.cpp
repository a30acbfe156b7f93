// One instance range over several GPUs behind the C ABI (include/oalsfx_hip.h, oalsfx_group_*): BASELINE configs[4] is 262 144 EAX
// reverbs over the eight GPUs of a node, and the reference's instances are independent of each other (src/oalsfxpp.cpp:2984-3037 touches
// nothing outside its own Api::Impl): the split is a contiguous range per device, one batch and one host thread per device, no
// collective and no peer traffic.  Built on the batch entry points only -- a group is what a C++ caller would otherwise write
// around N batches by hand.
//
// Threads: every device has a worker thread that lives as long as the group.  The calls that block per device (host-pointer mixes:
// copy in, kernels, copy out, wait) are fanned out to the workers and joined; the calls that only queue work (setters, device-buffer
// mixes) run on the caller's thread, device after device -- every batch entry point selects its batch's device for the calling thread.
// A group, like a batch and like the reference's Api, is not thread-safe; distinct groups are independent.
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "oalsfx_hip.h"

namespace {

struct Worker {
    std::thread thread;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> task; // (one at a time: the caller joins before it posts the next)
    bool has_task = false, done = false, quit = false;
    int result = 1;

    void loop()
    {
        std::unique_lock<std::mutex> lock(m);
        for (;;) {
            cv.wait(lock, [&] { return has_task || quit; });
            if (quit) return;
            std::function<int()> t = std::move(task);
            has_task = false;
            lock.unlock();
            const int r = t();
            lock.lock();
            result = r;
            done = true;
            cv.notify_all();
        }
    }
    void post(std::function<int()> t)
    {
        std::lock_guard<std::mutex> lock(m);
        task = std::move(t);
        has_task = true;
        done = false;
        cv.notify_all();
    }
    int join()
    {
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [&] { return done; });
        return result;
    }
};

} // namespace

struct oalsfx_group {
    int n = 0, channels = 0;
    std::vector<int> device, first, count; // per shard
    std::vector<oalsfx_batch*> batch;
    std::vector<Worker*> worker;
    std::string error;

    bool fail(const std::string& msg) { error = msg; return false; }
    bool shard_failed(size_t k, const char* what)
    {
        char head[96];
        std::snprintf(head, sizeof(head), "device %d (instances %d .. %d), %s: ", device[k], first[k], first[k] + count[k] - 1, what);
        error = std::string(head) + oalsfx_batch_error(batch[k]);
        return false;
    }
    // the part of the global range [f, f + c) that shard k owns: its first instance there and how many (0: none)
    int overlap(size_t k, int f, int c, int* local_first) const
    {
        const int lo = f > first[k] ? f : first[k], hi = (f + c < first[k] + count[k]) ? f + c : first[k] + count[k];
        *local_first = lo - first[k];
        return hi > lo ? hi - lo : 0;
    }
};

namespace {
std::string g_group_error;

// contiguous ranges whose sizes differ by at most one (oalsfxpp_amd/sharding.py: shard_range)
void shard_range(int n_total, int k, int shards, int* first, int* count)
{
    const int base = n_total / shards, extra = n_total % shards;
    *first = k * base + (k < extra ? k : extra);
    *count = base + (k < extra ? 1 : 0);
}
} // namespace

extern "C" {

const char* oalsfx_group_last_error(void) { return g_group_error.c_str(); }

oalsfx_group* oalsfx_group_create(int n_total, const int* device_ids, int n_devices, int channel_format, int sampling_rate, int effect_count)
{
    g_group_error.clear();
    if (n_devices <= 0 || !device_ids) { g_group_error = "A group needs at least one device."; return nullptr; }
    if (n_total < n_devices) { g_group_error = "Instance count is out of range (fewer instances than devices)."; return nullptr; }
    oalsfx_group* g = new oalsfx_group;
    g->n = n_total;
    for (int k = 0; k < n_devices; ++k) {
        int f = 0, c = 0;
        shard_range(n_total, k, n_devices, &f, &c);
        oalsfx_batch* b = oalsfx_batch_create(c, channel_format, sampling_rate, effect_count, device_ids[k]);
        if (!b) {
            char head[64];
            std::snprintf(head, sizeof(head), "device %d: ", device_ids[k]);
            g_group_error = std::string(head) + oalsfx_last_error();
            for (oalsfx_batch* made : g->batch) oalsfx_batch_destroy(made);
            delete g;
            return nullptr;
        }
        g->device.push_back(device_ids[k]);
        g->first.push_back(f);
        g->count.push_back(c);
        g->batch.push_back(b);
    }
    g->channels = oalsfx_batch_channels(g->batch[0]);
    for (int k = 0; k < n_devices; ++k) {
        Worker* w = new Worker;
        w->thread = std::thread([w] { w->loop(); });
        g->worker.push_back(w);
    }
    return g;
}

void oalsfx_group_destroy(oalsfx_group* g)
{
    if (!g) return;
    for (Worker* w : g->worker) {
        { std::lock_guard<std::mutex> lock(w->m); w->quit = true; w->cv.notify_all(); }
        w->thread.join();
        delete w;
    }
    for (oalsfx_batch* b : g->batch) oalsfx_batch_destroy(b);
    delete g;
}

const char* oalsfx_group_error(const oalsfx_group* g) { return g ? g->error.c_str() : g_group_error.c_str(); }
int oalsfx_group_instances(const oalsfx_group* g) { return g->n; }
int oalsfx_group_channels(const oalsfx_group* g) { return g->channels; }
int oalsfx_group_devices(const oalsfx_group* g) { return static_cast<int>(g->batch.size()); }

int oalsfx_group_shard(const oalsfx_group* g, int k, int* device_id, int* first, int* count)
{
    if (k < 0 || k >= static_cast<int>(g->batch.size())) return 0;
    if (device_id) *device_id = g->device[k];
    if (first) *first = g->first[k];
    if (count) *count = g->count[k];
    return 1;
}

oalsfx_batch* oalsfx_group_batch(oalsfx_group* g, int k) { return (k < 0 || k >= static_cast<int>(g->batch.size())) ? nullptr : g->batch[k]; }

// ---- setters over the global instance range: every shard gets its part ----
#define OALSFX_GROUP_RANGE_CHECK()                                                                          \
    if (first < 0 || count < 0 || first + count > g->n) return g->fail("Instance range is out of range.") ? 1 : 0

int oalsfx_group_set_effect(oalsfx_group* g, int first, int count, int slot, const oalsfx_effect* effects, int stride_bytes)
{
    OALSFX_GROUP_RANGE_CHECK();
    for (size_t k = 0; k < g->batch.size(); ++k) {
        int lf = 0;
        const int c = g->overlap(k, first, count, &lf);
        if (!c) continue;
        const auto* e = reinterpret_cast<const oalsfx_effect*>(reinterpret_cast<const char*>(effects) + static_cast<size_t>(g->first[k] + lf - first) * stride_bytes);
        if (!oalsfx_batch_set_effect(g->batch[k], lf, c, slot, e, stride_bytes)) return g->shard_failed(k, "set_effect") ? 1 : 0;
    }
    return 1;
}

int oalsfx_group_set_effect_type(oalsfx_group* g, int first, int count, int slot, int effect_type)
{
    OALSFX_GROUP_RANGE_CHECK();
    for (size_t k = 0; k < g->batch.size(); ++k) {
        int lf = 0;
        const int c = g->overlap(k, first, count, &lf);
        if (c && !oalsfx_batch_set_effect_type(g->batch[k], lf, c, slot, effect_type)) return g->shard_failed(k, "set_effect_type") ? 1 : 0;
    }
    return 1;
}

int oalsfx_group_set_effect_props(oalsfx_group* g, int first, int count, int slot, const void* props, int stride_bytes)
{
    OALSFX_GROUP_RANGE_CHECK();
    for (size_t k = 0; k < g->batch.size(); ++k) {
        int lf = 0;
        const int c = g->overlap(k, first, count, &lf);
        if (!c) continue;
        const void* p = static_cast<const char*>(props) + static_cast<size_t>(g->first[k] + lf - first) * stride_bytes;
        if (!oalsfx_batch_set_effect_props(g->batch[k], lf, c, slot, p, stride_bytes)) return g->shard_failed(k, "set_effect_props") ? 1 : 0;
    }
    return 1;
}

int oalsfx_group_set_send_props(oalsfx_group* g, int first, int count, int slot, const oalsfx_send_props* props)
{
    OALSFX_GROUP_RANGE_CHECK();
    for (size_t k = 0; k < g->batch.size(); ++k) {
        int lf = 0;
        const int c = g->overlap(k, first, count, &lf);
        if (c && !oalsfx_batch_set_send_props(g->batch[k], lf, c, slot, props)) return g->shard_failed(k, "set_send_props") ? 1 : 0;
    }
    return 1;
}

int oalsfx_group_apply_changes(oalsfx_group* g, int first, int count)
{
    OALSFX_GROUP_RANGE_CHECK();
    for (size_t k = 0; k < g->batch.size(); ++k) {
        int lf = 0;
        const int c = g->overlap(k, first, count, &lf);
        if (c && !oalsfx_batch_apply_changes(g->batch[k], lf, c)) return g->shard_failed(k, "apply_changes") ? 1 : 0;
    }
    return 1;
}

// ---- the hot path ----
// Host buffers of the whole range, [n_total][frames][channels]: every device's worker copies its range in, runs its kernels, copies
// out and waits; the call returns when all have (Api::mix for every instance of the group, reference src/oalsfxpp.cpp:3785-3829).
int oalsfx_group_mix(oalsfx_group* g, int frames, const float* src_host, float* dst_host)
{
    if (frames < 0) return g->fail("Frame count is out of range.") ? 1 : 0;
    if (frames == 0) return 1;
    if (!src_host || !dst_host) return g->fail(!src_host ? "Null source samples." : "Null target samples.") ? 1 : 0;
    const size_t per_instance = static_cast<size_t>(frames) * g->channels;
    for (size_t k = 0; k < g->batch.size(); ++k) {
        oalsfx_batch* b = g->batch[k];
        const float* s = src_host + static_cast<size_t>(g->first[k]) * per_instance;
        float* d = dst_host + static_cast<size_t>(g->first[k]) * per_instance;
        g->worker[k]->post([b, frames, s, d] { return oalsfx_batch_mix(b, frames, s, d); });
    }
    bool ok = true;
    for (size_t k = 0; k < g->batch.size(); ++k)
        if (!g->worker[k]->join() && ok) ok = g->shard_failed(k, "mix");
    return ok ? 1 : 0;
}

// Buffers resident on each device (src_per_device[k], dst_per_device[k]: that shard's [count][frames][channels]): queued on every
// batch's own stream, device after device, without waiting -- consecutive calls overlap on each device as they do for a batch alone.
int oalsfx_group_mix_device(oalsfx_group* g, int frames, const float* const* src_per_device, float* const* dst_per_device)
{
    if (!src_per_device || !dst_per_device) return g->fail("Null buffer table.") ? 1 : 0;
    for (size_t k = 0; k < g->batch.size(); ++k)
        if (!oalsfx_batch_mix_device(g->batch[k], frames, src_per_device[k], dst_per_device[k], nullptr)) return g->shard_failed(k, "mix_device") ? 1 : 0;
    return 1;
}

int oalsfx_group_synchronize(oalsfx_group* g)
{
    bool ok = true;
    for (size_t k = 0; k < g->batch.size(); ++k)
        if (!oalsfx_batch_synchronize(g->batch[k]) && ok) ok = g->shard_failed(k, "synchronize");
    return ok ? 1 : 0;
}

} // extern "C"
