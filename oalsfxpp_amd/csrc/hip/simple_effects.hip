// Process kernels for the ring-light effects: null, chorus, flanger, compressor, dedicated,
// distortion, echo, equalizer, ring modulator.
//
// Each replaces the matching EffectState::do_process body of the reference (line ranges at each
// struct below).  First version of these kernels: one lane per instance, samples in order, state in
// registers, so every serial recurrence rounds exactly like the reference by construction.  The
// per-sample accumulation order into an output channel is the reference's (left tap before right tap,
// B-format channel 0..3, ...); the reference's internal 64/128/256-sample chunking is invisible on a
// stream and is not reproduced.
#include "common.hpp"

namespace oalsfx_hip {

namespace {

// ---- per-effect sample steppers: init() loads parameters and state, step() consumes one B-format
// frame `wet[4]` and accumulates into out[CH], finish() stores the state ----

struct NullFx {
    __device__ void init(const oalsfx_slot_params&, oalsfx_slot_state&, float*) {}
    template <int CH> __device__ void step(const float*, float*, int) {}
    __device__ void finish(oalsfx_slot_state&) {}
};

// chorus / flanger (reference src/oalsfxpp.cpp:4113-4276, 5384-5547)
struct ModDelayFx {
    oalsfx_moddelay_params p;
    int offset;
    float* side[2];
    __device__ void init(const oalsfx_slot_params& sp, oalsfx_slot_state& ss, float* ring)
    {
        p = sp.u.moddelay;
        offset = ss.u.moddelay.offset;
        side[0] = ring;
        side[1] = ring + p.ring_len;
    }
    __device__ int lfo_delay(int phase) const
    {
        if (p.waveform == 1) return static_cast<int>((1.0F - fabsf(2.0F - (p.lfo_scale * phase))) * p.depth) + p.delay;
        return static_cast<int>(glibc_sinf(p.lfo_scale * phase) * p.depth) + p.delay;
    }
    template <int CH> __device__ void step(const float* wet, float* out, int channels)
    {
        const int mask = p.ring_len - 1;
        const int phase[2] = {offset % p.lfo_range, (offset + p.lfo_disp) % p.lfo_range};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float* buf = side[k];
            buf[offset & mask] = wet[0];
            const float t = buf[(offset - lfo_delay(phase[k])) & mask] * p.feedback;
            buf[offset & mask] += t;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const float g = p.gains[k][c];
                if (c < channels && audible(g)) out[c] += t * g;
            }
        }
        offset += 1;
    }
    __device__ void finish(oalsfx_slot_state& ss) { ss.u.moddelay.offset = offset; }
};

// compressor (reference src/oalsfxpp.cpp:4352-4453)
struct CompressorFx {
    oalsfx_compressor_params p;
    float gc;
    __device__ void init(const oalsfx_slot_params& sp, oalsfx_slot_state& ss, float*)
    {
        p = sp.u.compressor;
        gc = ss.u.compressor.gain_control;
    }
    template <int CH> __device__ void step(const float* wet, float* out, int channels)
    {
        float amplitude = 1.0F;
        if (p.enabled) {
            amplitude = fabsf(wet[0]);
            amplitude = fmaxf(amplitude + fabsf(wet[1]), fmaxf(amplitude + fabsf(wet[2]), amplitude + fabsf(wet[3])));
        }
        if (amplitude > gc) gc = fminf(gc + p.attack_rate, amplitude);
        else if (amplitude < gc) gc = fmaxf(gc - p.release_rate, amplitude);
        const float output = 1.0F / fminf(2.0F, fmaxf(0.5F, gc));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = wet[j] * output;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const float g = p.gains[j][c];
                if (c < channels && audible(g)) out[c] += g * t;
            }
        }
    }
    __device__ void finish(oalsfx_slot_state& ss) { ss.u.compressor.gain_control = gc; }
};

// dedicated dialog / LFE (reference src/oalsfxpp.cpp:4556-4576)
struct DedicatedFx {
    oalsfx_dedicated_params p;
    __device__ void init(const oalsfx_slot_params& sp, oalsfx_slot_state&, float*) { p = sp.u.dedicated; }
    template <int CH> __device__ void step(const float* wet, float* out, int channels)
    {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const float g = p.gains[c];
            if (c < channels && audible(g)) out[c] += wet[0] * g;
        }
    }
    __device__ void finish(oalsfx_slot_state&) {}
};

// distortion, 4x zero-stuffed oversampling (reference src/oalsfxpp.cpp:4675-4750)
struct DistortionFx {
    oalsfx_distortion_params p;
    oalsfx_hist_t lp, bp;
    __device__ void init(const oalsfx_slot_params& sp, oalsfx_slot_state& ss, float*)
    {
        p = sp.u.distortion;
        lp = ss.u.distortion.low_pass;
        bp = ss.u.distortion.band_pass;
    }
    template <int CH> __device__ void step(const float* wet, float* out, int channels)
    {
        const float fc = p.edge_coeff;
        float kept = 0.0F;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float in = (k == 0) ? wet[0] * 4.0F : 0.0F;
            float smp = biquad_step(p.low_pass, lp, in);
            smp = (1.0F + fc) * smp / (1.0F + (fc * fabsf(smp)));
            smp = (1.0F + fc) * smp / (1.0F + (fc * fabsf(smp))) * -1.0F;
            smp = (1.0F + fc) * smp / (1.0F + (fc * fabsf(smp)));
            const float y = biquad_step(p.band_pass, bp, smp);
            if (k == 0) kept = y;
        }
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const float g = p.gains[c] * p.attenuation;
            if (c < channels && audible(g)) out[c] += g * kept;
        }
    }
    __device__ void finish(oalsfx_slot_state& ss)
    {
        ss.u.distortion.low_pass = lp;
        ss.u.distortion.band_pass = bp;
    }
};

// echo (reference src/oalsfxpp.cpp:4887-4962)
struct EchoFx {
    oalsfx_echo_params p;
    oalsfx_hist_t h;
    int offset;
    float* ring;
    __device__ void init(const oalsfx_slot_params& sp, oalsfx_slot_state& ss, float* r)
    {
        p = sp.u.echo;
        h = ss.u.echo.filter;
        offset = ss.u.echo.offset;
        ring = r;
    }
    template <int CH> __device__ void step(const float* wet, float* out, int channels)
    {
        const int mask = p.ring_len - 1;
        const float t1 = ring[(offset - p.tap1) & mask];
        const float t2 = ring[(offset - p.tap2) & mask];
        const float in = t2 + wet[0];
        const float y = (in * p.filter.b0) + (h.x[0] * p.filter.b1) + (h.x[1] * p.filter.b2) - (h.y[0] * p.filter.a1) - (h.y[1] * p.filter.a2);
        h.x[1] = h.x[0]; h.x[0] = in;
        h.y[1] = h.y[0]; h.y[0] = y;
        ring[offset & mask] = y * p.feed_gain;
        offset += 1;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (c >= channels) continue;
            const float g0 = p.gains[0][c];
            if (audible(g0)) out[c] += t1 * g0;
            const float g1 = p.gains[1][c];
            if (audible(g1)) out[c] += t2 * g1;
        }
    }
    __device__ void finish(oalsfx_slot_state& ss)
    {
        ss.u.echo.filter = h;
        ss.u.echo.offset = offset;
    }
};

// equalizer (reference src/oalsfxpp.cpp:5161-5213)
struct EqualizerFx {
    oalsfx_equalizer_params p;
    oalsfx_equalizer_state s;
    __device__ void init(const oalsfx_slot_params& sp, oalsfx_slot_state& ss, float*)
    {
        p = sp.u.equalizer;
        s = ss.u.equalizer;
    }
    template <int CH> __device__ void step(const float* wet, float* out, int channels)
    {
#pragma unroll
        for (int ft = 0; ft < 4; ++ft) {
            float v = wet[ft];
#pragma unroll
            for (int b = 0; b < 4; ++b) v = biquad_step(p.band[b], s.hist[b][ft], v);
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const float g = p.gains[ft][c];
                if (c < channels && audible(g)) out[c] += g * v;
            }
        }
    }
    __device__ void finish(oalsfx_slot_state& ss) { ss.u.equalizer = s; }
};

// ring modulator (reference src/oalsfxpp.cpp:5652-5784)
struct RingModFx {
    oalsfx_ringmod_params p;
    oalsfx_ringmod_state s;
    __device__ void init(const oalsfx_slot_params& sp, oalsfx_slot_state& ss, float*)
    {
        p = sp.u.ringmod;
        s = ss.u.ringmod;
    }
    __device__ float carrier(int index) const
    {
        constexpr int frac_bits = 24;
        constexpr int frac_one = 1 << frac_bits;
        if (p.waveform == 0) return glibc_sinf(index * (6.28318530717958647692F / frac_one) - 3.14159265358979323846F) * 0.5F + 0.5F;
        if (p.waveform == 1) return static_cast<float>(index) / frac_one;
        return static_cast<float>((index >> (frac_bits - 1)) & 1);
    }
    template <int CH> __device__ void step(const float* wet, float* out, int channels)
    {
        s.index = (s.index + p.step) & ((1 << 24) - 1);
        const float m = carrier(s.index);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = biquad_step(p.filter, s.hist[j], wet[j]) * m;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const float g = p.gains[j][c];
                if (c < channels && audible(g)) out[c] += g * t;
            }
        }
    }
    __device__ void finish(oalsfx_slot_state& ss) { ss.u.ringmod = s; }
};

} // namespace

// One lane per instance.  The front end (dry mix / B-format send of mix_source, reference
// src/oalsfxpp.cpp:2917-2982) and the back end (write_f32, src/oalsfxpp.cpp:3414-3431) are fused in,
// selected by `flags`.
template <int CH, class Fx>
__global__ __launch_bounds__(64) void k_simple(KernelCtx ctx, int slot, const int* __restrict__ list, int count, int flags)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= count) return;
    const int inst = list[tid];
    const int channels = (CH == 8) ? ctx.channels : CH;
    const int frames = ctx.frames;
    const size_t sidx = static_cast<size_t>(inst) * ctx.slots + slot;
    const oalsfx_slot_params& SP = ctx.params[sidx];
    oalsfx_slot_state& SS = ctx.state[sidx];
    const oalsfx_source_params& SRC = ctx.source[inst];
    const bool first = (flags & kFirst) != 0;
    const bool last = (flags & kLast) != 0;

    const bool filtered = (flags & kFiltered) != 0;
    const float* src = ctx.src + static_cast<size_t>(inst) * ctx.src_stride;
    const float* wsrc = ctx.wet_src + static_cast<size_t>(inst) * ctx.src_stride;
    float* dst = ctx.dst + static_cast<size_t>(inst) * ctx.io_stride;
    float* mixbuf = ctx.mixbuf ? ctx.mixbuf + static_cast<size_t>(inst) * channels * OALSFX_MAX_CHUNK : nullptr;

    // send gains of this instance into registers
    float dry_gain[CH][CH];
    float aux_gain[CH][4];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
        for (int o = 0; o < CH; ++o) dry_gain[c][o] = (c < channels && o < channels) ? SRC.direct.gains[c][o] : 0.0F;
#pragma unroll
        for (int k = 0; k < 4; ++k) aux_gain[c][k] = (c < channels) ? SRC.aux[slot].gains[c][k] : 0.0F;
    }
    const bool send_on = SRC.aux[slot].out_channels != 0;

    Fx fx;
    fx.init(SP, SS, ctx.rings[sidx]);

    for (int i = 0; i < frames; ++i) {
        float in[CH], win[CH], out[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            in[c] = (c < channels) ? src[static_cast<size_t>(i) * channels + c] : 0.0F;
            win[c] = (filtered && c < channels) ? wsrc[static_cast<size_t>(i) * channels + c] : in[c];
            out[c] = 0.0F;
        }
        float wet[4] = {0.0F, 0.0F, 0.0F, 0.0F};
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (c >= channels) continue;
            if (first) {
#pragma unroll
                for (int o = 0; o < CH; ++o)
                    if (o < channels && audible(dry_gain[c][o])) out[o] += in[c] * dry_gain[c][o];
            }
            if (send_on) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (audible(aux_gain[c][k])) wet[k] += win[c] * aux_gain[c][k];
            }
        }
        if (!first) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (c < channels) out[c] = mixbuf[c * OALSFX_MAX_CHUNK + i];
        }

        fx.template step<CH>(wet, out, channels);

        if (last) {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (c < channels) dst[static_cast<size_t>(i) * channels + c] = out[c];
        } else {
#pragma unroll
            for (int c = 0; c < CH; ++c)
                if (c < channels) mixbuf[c * OALSFX_MAX_CHUNK + i] = out[c];
        }
    }

    fx.finish(SS);
    SS.seen_seq = SP.update_seq;

    if (first && !filtered)
        for (int c = 0; c < channels; ++c) send_history_follow(ctx, inst, c, channels, frames, src);
}

// Send shelf filters as a pre-pass (reference apply_filters, src/oalsfxpp.cpp:3101-3143, called from mix_source :2929-2965).
// One wavefront per instance; lane = send * 8 + input channel runs that send's two biquads over the chunk in sample order,
// so the recurrences round like the reference's.  Sends without a filter copy their input, which lets the effect kernels
// read every send from the same place.  Runs only while some instance of the batch has a filter switched on.
__global__ __launch_bounds__(256) void k_send_filters(KernelCtx ctx, const float* __restrict__ src_all, long long src_stride,
                                                      float* __restrict__ filtered, size_t send_floats, int instances)
{
    const int lane = threadIdx.x & 63;
    const int inst = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (inst >= instances) return;
    const int channels = ctx.channels;
    const int send = lane >> 3;
    const int c = lane & 7;
    if (send > ctx.slots || c >= channels) return;
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
    for (int base = 0; base < frames; base += 8) {
        float x[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = (base + k < frames) ? src[static_cast<size_t>(base + k) * channels] : 0.0F;
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
    hipLaunchKernelGGL(k_send_filters, dim3((instances + 3) / 4), dim3(256), 0, stream, ctx, src, src_stride, filtered, send_floats, instances);
}

template <class Fx>
static void launch_fx(const KernelCtx& ctx, int slot, const int* list, int count, int flags, hipStream_t stream)
{
    const dim3 grid((count + 63) / 64), block(64);
    if (ctx.channels == 1) hipLaunchKernelGGL((k_simple<1, Fx>), grid, block, 0, stream, ctx, slot, list, count, flags);
    else if (ctx.channels == 2) hipLaunchKernelGGL((k_simple<2, Fx>), grid, block, 0, stream, ctx, slot, list, count, flags);
    else hipLaunchKernelGGL((k_simple<8, Fx>), grid, block, 0, stream, ctx, slot, list, count, flags);
}

void launch_simple(int effect_type, const KernelCtx& ctx, int slot, const int* list, int count, int flags, hipStream_t stream)
{
    if (count <= 0) return;
    switch (effect_type) {
    case OALSFX_NULL: launch_fx<NullFx>(ctx, slot, list, count, flags, stream); break;
    case OALSFX_CHORUS:
    case OALSFX_FLANGER: launch_fx<ModDelayFx>(ctx, slot, list, count, flags, stream); break;
    case OALSFX_COMPRESSOR: launch_fx<CompressorFx>(ctx, slot, list, count, flags, stream); break;
    case OALSFX_DEDICATED_DIALOG:
    case OALSFX_DEDICATED_LFE: launch_fx<DedicatedFx>(ctx, slot, list, count, flags, stream); break;
    case OALSFX_DISTORTION: launch_fx<DistortionFx>(ctx, slot, list, count, flags, stream); break;
    case OALSFX_ECHO: launch_fx<EchoFx>(ctx, slot, list, count, flags, stream); break;
    case OALSFX_EQUALIZER: launch_fx<EqualizerFx>(ctx, slot, list, count, flags, stream); break;
    case OALSFX_RING_MODULATOR: launch_fx<RingModFx>(ctx, slot, list, count, flags, stream); break;
    default: break;
    }
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
