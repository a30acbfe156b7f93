// The ring-light effects' own kernel: one wavefront per listed instance (bodies in wave_effects_body.hpp).
#include "wave_effects_body.hpp"

namespace oalsfx_hip {

// One wavefront per listed instance, any mix of the ring-light effect types, for `slot_count` consecutive slots starting at
// `slot`: the host fuses runs of slots that hold no reverb at all, so that an instance's chorus -> flanger -> echo chain
// is one launch (the slots still accumulate in order, through mixbuf, by the same wavefront).
template <int CH>
__global__ __launch_bounds__(256) void k_wave_effects(KernelCtx ctx, int slot, int slot_count, const int* __restrict__ list, int count, int flags)
{
    __shared__ __attribute__((aligned(16))) float lds_all[4][wfx::kLdsFloats];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int w = blockIdx.x * 4 + wib;
    if (w >= count) return; // whole wavefronts leave; the kernel has no workgroup barrier
    const int inst = __builtin_amdgcn_readfirstlane(list[w]);
    wfx::wave_slots<CH>(ctx, slot, slot_count, inst, flags, lds_all[wib], lane);
}

void launch_wave_effects(const KernelCtx& ctx, int slot, int slot_count, const int* list, int count, int flags, hipStream_t stream)
{
    if (count <= 0 || slot_count <= 0) return;
    const dim3 grid((count + 3) / 4), block(256);
    if (ctx.channels == 1) OALSFX_LAUNCH((k_wave_effects<1>), grid, block, stream, ctx, slot, slot_count, list, count, flags);
    else if (ctx.channels == 2) OALSFX_LAUNCH((k_wave_effects<2>), grid, block, stream, ctx, slot, slot_count, list, count, flags);
    else OALSFX_LAUNCH((k_wave_effects<8>), grid, block, stream, ctx, slot, slot_count, list, count, flags);
}

} // namespace oalsfx_hip
