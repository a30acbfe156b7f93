// Kernels around the effect bodies: the send-filter pre-pass, the benchmark's synthetic input generator and two
// measurement helpers.  (The effect bodies live in reverb.hip and wave_effects.hip.)
#include "common.hpp"

namespace oalsfx_hip {

// Send shelf filters as a pre-pass (reference apply_filters, src/oalsfxpp.cpp:3101-3143, called from mix_source :2929-2965).
// One lane per (instance, send, input channel): it runs that send's two biquads over the chunk in sample order, so the
// recurrences round like the reference's; a wavefront packs as many instances as fit its 64 lanes (16 for a stereo batch
// with one slot).  Only instances with a filter switched on are handled (their sends without one copy their input, which
// lets the effect kernels read every send of such an instance from the same place).  Runs only while some instance of the
// batch has a filter switched on.
__global__ __launch_bounds__(256) void k_send_filters(KernelCtx ctx, const float* __restrict__ src_all, long long src_stride,
                                                      float* __restrict__ filtered, size_t send_floats, int instances)
{
    const int lane = threadIdx.x & 63;
    const int channels = ctx.channels;
    const int lanes_per_instance = (1 + ctx.slots) * channels;   // <= 5 * 8
    const int instances_per_wave = 64 / lanes_per_instance;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int sub = lane / lanes_per_instance, r = lane % lanes_per_instance;
    const int inst = wave * instances_per_wave + sub;
    if (sub >= instances_per_wave || inst >= instances) return;
    if (!instance_has_send_filter(ctx, inst)) return; // the effect kernels read the raw input for this instance and keep its histories
    const int send = r / channels;
    const int c = r % channels;
    const oalsfx_source_params& P = ctx.source[inst];
    const oalsfx_send_params& sp = send == 0 ? P.direct : P.aux[send - 1];
    if (send > 0 && sp.out_channels == 0) return; // null slot: the send is disabled and its history frozen
    oalsfx_source_state& S = ctx.source_state[inst];
    oalsfx_hist_t lp = S.lp[send][c];
    oalsfx_hist_t hp = S.hp[send][c];
    const oalsfx_biquad_t clp = sp.lp, chp = sp.hp;
    const int type = sp.filter_type;
    const float* src = src_all + static_cast<size_t>(inst) * src_stride + c;
    float* out = filtered + static_cast<size_t>(send) * send_floats + static_cast<size_t>(inst) * ctx.src_stride + c;
    const int frames = ctx.frames;
    // eight frames at a time, the next eight requested before the current eight are filtered: the recurrence never waits for
    // the strided input
    float nx[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) nx[k] = (k < frames) ? src[static_cast<size_t>(k) * channels] : 0.0F;
    for (int base = 0; base < frames; base += 8) {
        float x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = nx[k];
#pragma unroll
        for (int k = 0; k < 8; ++k) nx[k] = (base + 8 + k < frames) ? src[static_cast<size_t>(base + 8 + k) * channels] : 0.0F;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (base + k >= frames) break;
            float y = x[k];
            if (type & OALSFX_AF_LOW_PASS) y = biquad_step(clp, lp, y);
            else { lp.x[1] = lp.x[0]; lp.x[0] = y; lp.y[1] = lp.y[0]; lp.y[0] = y; }
            if (type & OALSFX_AF_HIGH_PASS) {
                // with only the second filter on, it sees the raw input and the first follows the raw input too
                y = biquad_step(chp, hp, y);
            } else { hp.x[1] = hp.x[0]; hp.x[0] = y; hp.y[1] = hp.y[0]; hp.y[0] = y; }
            out[static_cast<size_t>(base + k) * channels] = y;
        }
    }
    S.lp[send][c] = lp;
    S.hp[send][c] = hp;
}

void launch_send_filters(const KernelCtx& ctx, const float* src, long long src_stride, float* filtered, size_t send_floats, int instances,
                         hipStream_t stream)
{
    if (instances <= 0 || ctx.frames <= 0) return;
    const int instances_per_wave = 64 / ((1 + ctx.slots) * ctx.channels);
    const int waves = (instances + instances_per_wave - 1) / instances_per_wave;
    hipLaunchKernelGGL(k_send_filters, dim3((waves + 3) / 4), dim3(256), 0, stream, ctx, src, src_stride, filtered, send_floats, instances);
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

void launch_stream_pattern(float* slabs, int instances, int dwords_per_lane, unsigned pos0, size_t slab_floats, int pos_skew, hipStream_t stream)
{
    const dim3 grid((instances + 3) / 4), block(256);
    if (dwords_per_lane == 1) hipLaunchKernelGGL(k_stream_pattern<1>, grid, block, 0, stream, slabs, instances, pos0, slab_floats, pos_skew);
    else if (dwords_per_lane == 2) hipLaunchKernelGGL(k_stream_pattern<2>, grid, block, 0, stream, slabs, instances, pos0, slab_floats, pos_skew);
    else hipLaunchKernelGGL(k_stream_pattern<4>, grid, block, 0, stream, slabs, instances, pos0, slab_floats, pos_skew);
}

} // namespace oalsfx_hip
