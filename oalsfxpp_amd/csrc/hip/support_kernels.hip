// Kernels around the effect bodies: the send-filter pre-pass, the benchmark's synthetic input generator and two
// measurement helpers.  (The effect bodies live in reverb.hip and wave_effects.hip.)
#include <algorithm>

#include "common.hpp"

namespace oalsfx_hip {

namespace { thread_local int t_lds_per_workgroup = 0; }
void set_lds_per_workgroup(int bytes) { t_lds_per_workgroup = bytes; }
int lds_per_workgroup() { return t_lds_per_workgroup; }

// Send shelf filters as a pre-pass (reference apply_filters, src/oalsfxpp.cpp:3101-3143, called from mix_source :2929-2965).
// A wavefront takes two consecutive instances (one where its lanes or LDS rows do not reach: a recurrence per send and input
// channel); the instances without a filter switched on are skipped -- the effect kernels read the raw input for them and keep
// their histories -- and a wavefront without any leaves at once.  Per 64-frame tile, with LDS rows [instance][input channel]
// and [instance][send][channel]:
//   1. lane = frame: the frames of every instance go into the input rows (requested one tile ahead);
//   2. lane = frame: per send, the feed-forward sums of the first shelf (or the input itself where that shelf is off);
//   3. lane = (instance, send, channel): the recurrence of the first shelf, in 16-sample register blocks;
//   4. lane = frame: the feed-forward sums of the second shelf, 5. its recurrence; 6. lane = frame: one plane per send out.
// A stage that is switched off passes its input through and lets its history follow it, as process_pass_through does
// (:1038-1056).  The histories stay in the recurrence lanes' registers from the first tile to the last.  A send to a null
// slot is disabled: no plane written, histories frozen (:2952-2956).
// (Tried before: whole filters, feed-forward sums included, in the recurrence lanes -- 24 instructions per
// sample on the serial path instead of 8.)
namespace {
constexpr int kFilterRow = 4 + 64; // [2], [3]: the two samples before the tile; data from [4], 16-byte aligned
constexpr int kFilterRowsPerWave = 48;
constexpr int kFilterInstances = 2; // per wavefront, see filter_instances_per_wave
constexpr int kFilterTable = kFilterInstances * 5 * 8; // per (instance, send): b0 b1 b2 of either shelf, which shelves are on, enabled

// instances per wavefront: two where they have lanes and LDS rows.  (Measured on the headline batch with a shelf on every
// direct send, step time against 59 us without filters: 1 instance per wavefront 90 us, 2: 78, 4: 86, 8: 106 -- the
// recurrences cost a wavefront the same for one lane or sixty-four, the lane = frame phases grow with the instances.)
inline __host__ __device__ int filter_instances_per_wave(int channels, int slots)
{
    const int chains = (1 + slots) * channels, rows = channels + chains;
    int n = 64 / chains;
    if (n > kFilterRowsPerWave / rows) n = kFilterRowsPerWave / rows;
    return n < 1 ? 1 : (n > kFilterInstances ? kFilterInstances : n);
}

// y = (u - a1 y1) - a2 y2 over n entries of a row that holds the feed-forward sums; outputs replace them
__device__ __forceinline__ void filter_recurrence(float* row, int n, float a1, float a2, float& y1, float& y2)
{
    auto step4 = [&](float4& v) {
        v.x = (v.x - (a1 * y1)) - (a2 * y2);
        v.y = (v.y - (a1 * v.x)) - (a2 * y1);
        v.z = (v.z - (a1 * v.y)) - (a2 * v.x);
        v.w = (v.w - (a1 * v.z)) - (a2 * v.y);
        y2 = v.z;
        y1 = v.w;
    };
    int i = 0;
    if (n == 64) {
        float4* r4 = reinterpret_cast<float4*>(row + 4);
        float4 c0 = r4[0], c1 = r4[1], c2 = r4[2], c3 = r4[3];
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
            float4 n0 = c0, n1 = c1, n2 = c2, n3 = c3;
            if (q + 4 < 16) { n0 = r4[q + 4]; n1 = r4[q + 5]; n2 = r4[q + 6]; n3 = r4[q + 7]; }
            step4(c0); step4(c1); step4(c2); step4(c3);
            r4[q] = c0; r4[q + 1] = c1; r4[q + 2] = c2; r4[q + 3] = c3;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        }
        i = 64;
    }
    for (; i < n; ++i) {
        const float y = (row[4 + i] - (a1 * y1)) - (a2 * y2);
        row[4 + i] = y;
        y2 = y1;
        y1 = y;
    }
}
}

__global__ __launch_bounds__(256) void k_send_filters(KernelCtx ctx, const float* __restrict__ src_all, long long src_stride,
                                                      float* __restrict__ filtered, size_t send_floats, const int* __restrict__ list, int instances)
{
    __shared__ __attribute__((aligned(16))) float filter_lds[4][kFilterRowsPerWave * kFilterRow + kFilterTable];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int channels = ctx.channels;
    const int sends = 1 + ctx.slots;
    const int chains = sends * channels; // <= 5 * 8
    const int rows = channels + chains;  // LDS rows of one instance: its input channels, then one row per recurrence
    const int ipw = filter_instances_per_wave(channels, ctx.slots);
    const int first = (blockIdx.x * 4 + wib) * ipw;
    if (first >= instances) return; // whole wavefronts leave; no workgroup barrier in this kernel
    // the wavefront's instances: entries first, first + 1, ... of the list (or the instances of those numbers), and which of them have a
    // filter switched on
    auto entry = [&](int k) -> int { return first + k < instances ? (list ? list[first + k] : first + k) : 0; };
    const int inst_of[kFilterInstances] = {__builtin_amdgcn_readfirstlane(entry(0)), __builtin_amdgcn_readfirstlane(entry(1))};
    const unsigned long long todo_mask = __ballot(lane < ipw && first + lane < instances && instance_has_send_filter(ctx, lane == 0 ? inst_of[0] : inst_of[1]));
    if (todo_mask == 0ULL) return;
    float* lds = filter_lds[wib];
    float* table = lds + kFilterRowsPerWave * kFilterRow;
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };

    // ---- recurrence lanes: (instance of the wavefront, send, input channel) ----
    const int sub = lane / chains, r = lane % chains;
    const int send = r / channels, c = r % channels;
    const bool chain_lane = sub < ipw && ((todo_mask >> sub) & 1ULL) != 0;
    const int my_inst = (chain_lane && sub == 1) ? inst_of[1] : inst_of[0];
    const oalsfx_source_params& P = ctx.source[my_inst];
    oalsfx_source_state& S = ctx.source_state[my_inst];
    const oalsfx_send_params* spp = &P.direct;
    if (chain_lane && send > 0) spp = &P.aux[send - 1];
    const oalsfx_send_params& sp = *spp;
    const bool enabled = chain_lane && (send == 0 || sp.out_channels != 0);
    const bool lp_on = (sp.filter_type & OALSFX_AF_LOW_PASS) != 0, hp_on = (sp.filter_type & OALSFX_AF_HIGH_PASS) != 0;
    const oalsfx_biquad_t clp = sp.lp, chp = sp.hp;
    // x: input, y: output of the first shelf = input of the second (the reference keeps that history twice; the copies are
    // equal after every call), z: output of the second shelf
    float x1 = 0.0F, x2 = 0.0F, y1 = 0.0F, y2 = 0.0F, z1 = 0.0F, z2 = 0.0F;
    if (enabled) {
        const oalsfx_hist_t lp = S.lp[send][c], hp = S.hp[send][c];
        x1 = lp.x[0]; x2 = lp.x[1]; y1 = lp.y[0]; y2 = lp.y[1]; z1 = hp.y[0]; z2 = hp.y[1];
    }
    if (chain_lane && c == 0) {
        // what the lane = frame phases need to know about this (instance, send)
        float* t = table + (sub * sends + send) * 8;
        t[0] = clp.b0; t[1] = clp.b1; t[2] = clp.b2; t[3] = chp.b0; t[4] = chp.b1; t[5] = chp.b2;
        reinterpret_cast<int*>(t)[6] = enabled ? ((lp_on ? 1 : 0) | (hp_on ? 2 : 0) | 4) : 0;
    }
    float* const wrow = lds + (sub * rows + channels + r) * kFilterRow; // this lane's recurrence row
    const float* const xrow = lds + (sub * rows + c) * kFilterRow;      // ... and its input channel

    const int frames = ctx.frames;
    // The frames of a tile are requested one tile ahead, all instances of the wavefront at once (mono / stereo: registers;
    // more channels: one instance per wavefront or two, read in place).
    float2 ahead[kFilterInstances];
    auto request = [&](int base) {
#pragma unroll
        for (int k = 0; k < kFilterInstances; ++k) {
            ahead[k] = make_float2(0.0F, 0.0F);
            if (k >= ipw || !((todo_mask >> k) & 1ULL) || base + lane >= frames) continue;
            const float* src = src_all + static_cast<size_t>(inst_of[k]) * src_stride;
            const size_t f = static_cast<size_t>(base + lane);
            if (channels == 2) ahead[k] = *reinterpret_cast<const float2*>(src + f * 2);
            else if (channels == 1) ahead[k].x = src[f];
        }
    };
    request(0);
    wave_sync(); // the table is in place
    for (int base = 0; base < frames; base += 64) {
        const int L = min(64, frames - base);
        // ---- 1. the tile's frames, lane = frame ----
        if (channels <= 2) {
#pragma unroll
            for (int k = 0; k < kFilterInstances; ++k) {
                if (k >= ipw || !((todo_mask >> k) & 1ULL) || lane >= L) continue;
                float* in_rows = lds + k * rows * kFilterRow;
                in_rows[4 + lane] = ahead[k].x;
                if (channels == 2) in_rows[kFilterRow + 4 + lane] = ahead[k].y;
            }
            if (base + 64 < frames) request(base + 64);
        } else {
            for (int k = 0; k < ipw; ++k) {
                if (!((todo_mask >> k) & 1ULL) || lane >= L) continue;
                const float* src = src_all + static_cast<size_t>((k == 1 ? inst_of[1] : inst_of[0])) * src_stride;
                float* in_rows = lds + k * rows * kFilterRow;
                const size_t f = static_cast<size_t>(base + lane);
                for (int ch = 0; ch < channels; ++ch) in_rows[ch * kFilterRow + 4 + lane] = src[f * channels + ch];
            }
        }
        wave_sync();
        // ---- 2. first shelf, feed-forward sums (lane = frame; samples 0 and 1 need the send's own input history, which the
        // recurrence lane holds: it writes them below) ----
        for (int k = 0; k < ipw; ++k) {
            if (!((todo_mask >> k) & 1ULL)) continue;
            for (int sd = 0; sd < sends; ++sd) {
                const float* t = table + (k * sends + sd) * 8;
                const int mode = reinterpret_cast<const int*>(t)[6];
                if (!(mode & 4)) continue;
                const float b0 = t[0], b1 = t[1], b2 = t[2];
                for (int ch = 0; ch < channels; ++ch) {
                    const float* x = lds + (k * rows + ch) * kFilterRow + 4 + lane;
                    float v = x[0];
                    if ((mode & 1) && lane >= 2) v = (b0 * x[0]) + (b1 * x[-1]) + (b2 * x[-2]);
                    if (lane < L) lds[(k * rows + channels + sd * channels + ch) * kFilterRow + 4 + lane] = v;
                }
            }
        }
        wave_sync();
        // ---- 3. first shelf, recurrence ----
        if (enabled) {
            const float in0 = xrow[4], in1 = xrow[5];
            if (lp_on) {
                wrow[4] = (clp.b0 * in0) + (clp.b1 * x1) + (clp.b2 * x2);
                if (L > 1) wrow[5] = (clp.b0 * in1) + (clp.b1 * in0) + (clp.b2 * x1);
            }
            wrow[2] = y2; wrow[3] = y1; // the second shelf's feed-forward sums read its input history here
            if (lp_on) {
                filter_recurrence(wrow, L, clp.a1, clp.a2, y1, y2);
            } else if (L >= 2) {
                y1 = xrow[4 + L - 1]; y2 = xrow[4 + L - 2];
            } else {
                y2 = y1; y1 = in0;
            }
            if (L >= 2) { x1 = xrow[4 + L - 1]; x2 = xrow[4 + L - 2]; }
            else { x2 = x1; x1 = in0; }
        }
        wave_sync();
        // ---- 4. second shelf, feed-forward sums (lane = frame): read, then replace ----
        for (int k = 0; k < ipw; ++k) {
            if (!((todo_mask >> k) & 1ULL)) continue;
            for (int sd = 0; sd < sends; ++sd) {
                const float* t = table + (k * sends + sd) * 8;
                const int mode = reinterpret_cast<const int*>(t)[6];
                if ((mode & 6) != 6) continue; // disabled, or the second shelf is off: the row already holds its output
                const float b0 = t[3], b1 = t[4], b2 = t[5];
                for (int ch = 0; ch < channels; ++ch) {
                    float* y = lds + (k * rows + channels + sd * channels + ch) * kFilterRow + 4 + lane;
                    const float v = (b0 * y[0]) + (b1 * y[-1]) + (b2 * y[-2]);
                    wave_sync(); // every lane has read its three samples
                    if (lane < L) y[0] = v;
                }
            }
        }
        wave_sync();
        // ---- 5. second shelf, recurrence ----
        if (enabled) {
            if (hp_on) filter_recurrence(wrow, L, chp.a1, chp.a2, z1, z2);
            else if (L >= 2) { z1 = y1; z2 = y2; }
            else { z2 = z1; z1 = y1; }
        }
        wave_sync();
        // ---- 6. the filtered planes, lane = frame ----
        for (int k = 0; k < ipw; ++k) {
            if (!((todo_mask >> k) & 1ULL) || lane >= L) continue;
            const int inst = (k == 1 ? inst_of[1] : inst_of[0]);
            const size_t f = static_cast<size_t>(base + lane);
            for (int sd = 0; sd < sends; ++sd) {
                if (!(reinterpret_cast<const int*>(table + (k * sends + sd) * 8)[6] & 4)) continue;
                float* out = filtered + static_cast<size_t>(sd) * send_floats + static_cast<size_t>(inst) * ctx.src_stride;
                const float* zr = lds + (k * rows + channels + sd * channels) * kFilterRow + 4 + lane;
                if (channels == 2) {
                    *reinterpret_cast<float2*>(out + f * 2) = make_float2(zr[0], zr[kFilterRow]);
                } else {
                    for (int ch = 0; ch < channels; ++ch) out[f * channels + ch] = zr[ch * kFilterRow];
                }
            }
        }
        wave_sync();
    }
    if (enabled) {
        oalsfx_hist_t lp, hp;
        lp.x[0] = x1; lp.x[1] = x2; lp.y[0] = y1; lp.y[1] = y2;
        hp.x[0] = y1; hp.x[1] = y2; hp.y[0] = z1; hp.y[1] = z2;
        S.lp[send][c] = lp;
        S.hp[send][c] = hp;
    }
}

// Chained launches (batch.cpp): a launch whose workgroups wait for the launch before must not take the chip before that
// launch has its workgroups on it; every chained launch but a run's first comes behind this gate.  (In the steady state of a run it
// finds its count reached: the launch before has been taking over the places of the launch two before, which has just completed.)
// (A gate that gives up leaves a mark beside the count, started[2], and the gates behind it do not wait at all: whatever held the launch
// before back -- a tool that runs kernels one at a time out of queue order -- will hold the next one back as well, and a second apiece
// adds up, a run of a thousand calls before the host next looks.  The host turns chaining off for the batch when it does: check_fault.)
__device__ __forceinline__ void chain_gate_wait(const unsigned* started, unsigned target, unsigned* fault)
{
    unsigned* gave_up = const_cast<unsigned*>(started) + 2;
    if (__hip_atomic_load(gave_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
    unsigned spins = 0;
    while (static_cast<int>(__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 22)) {
            if (fault) __hip_atomic_fetch_add(fault, kFaultGate, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

__global__ __launch_bounds__(64) void k_chain_gate(const unsigned* started, unsigned target, unsigned* fault)
{
    if (threadIdx.x == 0) chain_gate_wait(started, target, fault);
}

void launch_chain_gate(const unsigned* started, unsigned target, unsigned* fault, hipStream_t stream)
{
    hipLaunchKernelGGL(k_chain_gate, dim3(1), dim3(64), 0, stream, started, target, fault);
}

void launch_send_filters(const KernelCtx& ctx, const float* src, long long src_stride, float* filtered, size_t send_floats, const int* list, int instances,
                         hipStream_t stream)
{
    if (instances <= 0 || ctx.frames <= 0) return;
    const int ipw = filter_instances_per_wave(ctx.channels, ctx.slots);
    const int waves = (instances + ipw - 1) / ipw;
    hipLaunchKernelGGL(k_send_filters, dim3((waves + 3) / 4), dim3(256), 0, stream, ctx, src, src_stride, filtered, send_floats, list, instances);
}

// ---- synthetic benchmark input, generated in device memory ----
__global__ void k_fill_synthetic(float* dst, int instances, int floats_per_instance, unsigned buffer_index)
{
    const int inst = blockIdx.x * blockDim.x + threadIdx.x;
    if (inst >= instances) return;
    uint32_t x = synth_seed(static_cast<uint32_t>(inst), buffer_index);
    float* out = dst + static_cast<size_t>(inst) * floats_per_instance;
    for (int i = 0; i < floats_per_instance; ++i) {
        x ^= x << 13; x ^= x >> 17; x ^= x << 5;
        out[i] = static_cast<float>(x >> 8) * (1.0F / 8388608.0F) - 1.0F;
    }
}

void launch_fill_synthetic(float* dst, int instances, int floats_per_instance, unsigned buffer_index, hipStream_t stream)
{
    if (instances <= 0) return;
    hipLaunchKernelGGL(k_fill_synthetic, dim3((instances + 63) / 64), dim3(64), 0, stream, dst, instances, floats_per_instance, buffer_index);
}

// ---- parameter uploads: record k of a packed array goes to slot indices[k] of a device array (one 64-lane group per record)
__global__ __launch_bounds__(64) void k_scatter_records(unsigned* __restrict__ dst, int record_dwords, const unsigned* __restrict__ packed,
                                                        const int* __restrict__ indices, int count)
{
    const int k = blockIdx.x;
    if (k >= count) return;
    const size_t to = static_cast<size_t>(indices[k]) * record_dwords, from = static_cast<size_t>(k) * record_dwords;
    for (int i = threadIdx.x; i < record_dwords; i += 64) dst[to + i] = packed[from + i];
}

void launch_scatter_records(void* dst, size_t record_bytes, const void* packed, const int* indices, int count, hipStream_t stream)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(k_scatter_records, dim3(count), dim3(64), 0, stream, static_cast<unsigned*>(dst), static_cast<int>(record_bytes / 4),
                       static_cast<const unsigned*>(packed), indices, count);
}

// ---- the same for everything a parameter upload consists of, in one launch: up to four scatters and two plain copies.  The sources
// may lie in page-locked host memory (small uploads are read from the staging buffer directly): one launch then costs one round of
// reads over the link instead of one per array.
// Every load of a block is issued before its first store: the sources are read over the link (about 2 us a round trip), and a loop
// of load-store pairs would make one round trip per iteration (round 2's kernel: 10 us for four changed reverbs; this one 3).
__global__ __launch_bounds__(256) void k_upload(UploadJobs jobs)
{
    int k = blockIdx.x;
    const int t = threadIdx.x;
    // (the gate of the chained launch behind this upload: one wavefront's worth of waiting, beside the work of the others)
    if (blockIdx.x == 0 && t == 255 && jobs.gate_started != nullptr) chain_gate_wait(jobs.gate_started, jobs.gate_target, jobs.fault ? jobs.fault + 1 : nullptr); // (gates that give up count in the word behind the fault word)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const ScatterJob& sj = jobs.scatter[j];
        if (k < sj.count) {
            const unsigned* __restrict__ from = sj.packed + static_cast<size_t>(k) * sj.record_dwords;
            const int n = sj.record_dwords;
            // records of up to 1024 dwords in one round (the descriptors are 142 to 380), longer ones in further rounds
            for (int base = 0; base < n; base += 1024) {
                unsigned v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = base + t + 256 * r < n ? from[base + t + 256 * r] : 0u;
                const size_t to = static_cast<size_t>(sj.indices[k]) * n; // (read beside the record, not in front of it)
                if (base == 0 && sj.takes_turns && jobs.turn != nullptr && jobs.turn_wait != 0u) {
                    // chained launches: the launch before this upload's may still be at work on the instance, with the record as it is
                    // (a wait that counts out leaves the record alone: that launch is still reading it; the fault word fails the batch's
                    // next synchronising call and every call after it)
                    __shared__ int lost;
                    if (t == 0) {
                        unsigned spins = 0;
                        lost = 0;
                        while (__hip_atomic_load(jobs.turn + sj.indices[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != jobs.turn_wait) {
                            __builtin_amdgcn_s_sleep(2);
                            if (++spins > (1u << 20)) {
                                if (jobs.fault) __hip_atomic_fetch_add(jobs.fault, kFaultTurn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                lost = 1;
                                break;
                            }
                        }
                    }
                    __syncthreads();
                    if (lost) return;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (base + t + 256 * r < n) sj.dst[to + base + t + 256 * r] = v[r];
            }
            return;
        }
        k -= sj.count;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const CopyJob& cj = jobs.copy[j];
        if (k < cj.blocks) {
            // 1024 dwords per block: 16 bytes per lane, one round
            const size_t i = static_cast<size_t>(k) * 1024 + 4 * t;
            if (i + 4 <= cj.dwords) *reinterpret_cast<uint4*>(cj.dst + i) = *reinterpret_cast<const uint4*>(cj.src + i);
            else for (size_t q = i; q < cj.dwords; ++q) cj.dst[q] = cj.src[q];
            return;
        }
        k -= cj.blocks;
    }
}

void launch_upload(UploadJobs jobs, hipStream_t stream)
{
    int blocks = 0;
    for (auto& sj : jobs.scatter) blocks += sj.count > 0 ? sj.count : (sj.count = 0);
    for (auto& cj : jobs.copy) { cj.blocks = static_cast<int>((cj.dwords + 1023) / 1024); blocks += cj.blocks; }
    if (blocks == 0) return;
    hipLaunchKernelGGL(k_upload, dim3(blocks), dim3(256), 0, stream, jobs);
}

// ---- buffer copies between page-locked host memory and device memory as a kernel (oalsfx_batch_mix_async): the copy engines of some
// hosts run the two directions far below the link's rate when both are busy; a grid of wavefronts reading or writing the mapped host
// buffer does not depend on them.  16 bytes per lane, grid-stride.
__global__ __launch_bounds__(256) void k_copy_floats(float* __restrict__ dst, const float* __restrict__ src, size_t floats)
{
    const size_t quads = floats >> 2;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < quads; i += stride) d4[i] = s4[i];
    for (size_t i = (quads << 2) + static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < floats; i += stride) dst[i] = src[i];
}

void launch_copy_floats(float* dst, const float* src, size_t floats, hipStream_t stream)
{
    if (floats == 0) return;
    const int blocks = static_cast<int>(std::min<size_t>(128, (floats / 4 + 255) / 256 + 1)); // a few dozen wavefronts saturate the link; leave the chip to the effects
    hipLaunchKernelGGL(k_copy_floats, dim3(blocks), dim3(256), 0, stream, dst, src, floats);
}

// ---- an empty kernel: what an event pair around a launch measures beyond the kernel itself ----
__global__ void k_null() {}

void launch_null(hipStream_t stream) { hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, stream); }

// ---- HBM counter calibration (measurement helper): reads or writes a buffer with the access shape of the reverb
// kernel's ring traffic: one dword per lane, 256 contiguous bytes per wave instruction.  Run under
// rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE it gives the factor between counter values and real bytes for this shape.
__global__ void k_hbm_sweep(float* buf, size_t floats, int write, float* sink)
{
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    float acc = 0.0F;
    for (size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < floats; i += stride) {
        if (write) buf[i] = static_cast<float>(i & 1023);
        else acc += buf[i];
    }
    if (!write && acc == 12345.678F) sink[0] = acc; // keeps the loads alive
}

void launch_hbm_sweep(float* buf, size_t floats, int write, float* sink, hipStream_t stream)
{
    hipLaunchKernelGGL(k_hbm_sweep, dim3(256 * 8), dim3(256), 0, stream, buf, floats, write, sink);
}


// ---- measurement helper: the ring traffic of the steady-state reverb kernel without its arithmetic.  One wavefront per
// "instance" (a 942 080-byte slab like a 48 kHz reverb), 24 read streams at unaligned positions and 24 aligned write
// streams, 256 frames per launch, V consecutive dwords per lane and stream (V = 1: 256-byte bursts like the kernel today,
// 2: 512 bytes, 4: 1 KiB).  Shows what the memory system sustains for this pattern and what longer bursts would buy.
template <int V>
__global__ __launch_bounds__(256) void k_stream_pattern(float* slabs, int instances, unsigned pos0, size_t slab_floats, int pos_skew)
{
    const int lane = threadIdx.x & 63;
    const int inst = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (inst >= instances) return;
    float* slab = slabs + static_cast<size_t>(inst) * slab_floats;
    pos0 += static_cast<unsigned>(inst * pos_skew); // instances at different ring positions
    constexpr int kSteps = 256 / (64 * V);
    typedef float vf __attribute__((ext_vector_type(V)));
    vf cur[24];
    auto issue = [&](int step, vf* dst) {
#pragma unroll
        for (int s = 0; s < 24; ++s) {
            const unsigned p = (pos0 + step * 64 * V + lane * V + 13u + 37u * s) & 4095u; // unaligned, differs per stream
            const float* q = slab + s * 4608 + p;
#pragma unroll
            for (int k = 0; k < V; ++k) dst[s][k] = q[k]; // V consecutive dwords (not 16-byte aligned: as separate dwords)
        }
    };
    issue(0, cur);
    for (int step = 0; step < kSteps; ++step) {
        vf nxt[24];
        if (step + 1 < kSteps) issue(step + 1, nxt);
        __builtin_amdgcn_sched_barrier(0);
        vf acc = cur[0];
#pragma unroll
        for (int s = 1; s < 24; ++s) acc = acc + cur[s];
#pragma unroll
        for (int s = 0; s < 24; ++s) {
            const unsigned p = (pos0 + step * 64 * V + lane * V) & 4095u; // aligned like the kernel's writes
            vf* q = reinterpret_cast<vf*>(slab + 110592 + s * 4608 + p);
            *q = acc + static_cast<float>(s);
        }
        if (step + 1 < kSteps) {
#pragma unroll
            for (int s = 0; s < 24; ++s) cur[s] = nxt[s];
        }
    }
}

// ---- placement probe (batch.cpp: place_ring_chunk): the shape of the reverb kernel's ring traffic -- per slab 24 read streams at
// unaligned positions and 24 aligned write streams, 256 frames, one wavefront per slab, the next tile's reads requested ahead -- on a
// freshly zeroed chunk.  It writes the sums of what it read, i.e. zeros: the chunk stays zero-filled.  Where a chunk lands in the
// card's memory moves this traffic's rate by 15 % (profiles/README.md, vram map); the runtime keeps the fastest of a few candidates.
__global__ __launch_bounds__(256) void k_ring_probe(float* slabs, int instances, size_t slab_floats, unsigned pos0, int waves_per_slab)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int inst = wave / waves_per_slab; // several wavefronts per slab (at different positions) make a small chunk carry a full load
    if (inst >= instances) return;
    pos0 += static_cast<unsigned>(wave % waves_per_slab) * 1088u;
    float* slab = slabs + static_cast<size_t>(inst) * slab_floats;
    const unsigned span = static_cast<unsigned>(slab_floats / 48) & ~63u; // floats per stream region
    const unsigned wrap = span - 64u;
    float cur[24];
    auto issue = [&](int step, float* dst) {
#pragma unroll
        for (int s = 0; s < 24; ++s) dst[s] = slab[s * span + (pos0 + step * 64 + lane + 13u + 37u * s) % wrap];
    };
    issue(0, cur);
    for (int step = 0; step < 4; ++step) {
        float nxt[24];
        if (step + 1 < 4) issue(step + 1, nxt);
        __builtin_amdgcn_sched_barrier(0);
        float acc = cur[0];
#pragma unroll
        for (int s = 1; s < 24; ++s) acc += cur[s];
#pragma unroll
        for (int s = 0; s < 24; ++s) slab[(24 + s) * span + ((pos0 + step * 64) % wrap & ~63u) + lane] = acc;
        if (step + 1 < 4) {
#pragma unroll
            for (int s = 0; s < 24; ++s) cur[s] = nxt[s];
        }
    }
}

void launch_ring_probe(float* slabs, int instances, size_t slab_floats, unsigned pos0, int waves_per_slab, hipStream_t stream)
{
    const int waves = instances * waves_per_slab;
    hipLaunchKernelGGL(k_ring_probe, dim3((waves + 3) / 4), dim3(256), 0, stream, slabs, instances, slab_floats, pos0, waves_per_slab);
}

void launch_stream_pattern(float* slabs, int instances, int dwords_per_lane, unsigned pos0, size_t slab_floats, int pos_skew, hipStream_t stream)
{
    const dim3 grid((instances + 3) / 4), block(256);
    if (dwords_per_lane == 1) hipLaunchKernelGGL(k_stream_pattern<1>, grid, block, 0, stream, slabs, instances, pos0, slab_floats, pos_skew);
    else if (dwords_per_lane == 2) hipLaunchKernelGGL(k_stream_pattern<2>, grid, block, 0, stream, slabs, instances, pos0, slab_floats, pos_skew);
    else hipLaunchKernelGGL(k_stream_pattern<4>, grid, block, 0, stream, slabs, instances, pos0, slab_floats, pos_skew);
}

} // namespace oalsfx_hip
