// The ring-light effects' own kernel: one wavefront per listed instance (bodies in wave_effects_body.hpp).
#include "wave_effects_body.hpp"

namespace oalsfx_hip {

// One wavefront per listed instance, any mix of the ring-light effect types, for `slot_count` consecutive slots starting at
// `slot`: the host fuses runs of slots that hold no reverb at all, so that an instance's chorus -> flanger -> echo chain
// is one launch (the slots still accumulate in order, through mixbuf, by the same wavefront).
// 4096 instances are 1024 workgroups, four per compute unit: one round of the chip only while a wavefront keeps to 128 registers
// (at 133 the same grid took a second round and every effect type was 1.5 - 1.7 times slower), hence the occupancy bound.
template <int CH, bool CHN>
__global__ __launch_bounds__(256, CH == 8 ? 2 : 4) void k_wave_effects(KernelCtx ctx, int slot, int slot_count, const int* __restrict__ list, int count, WaveSegments seg,
                                                      int flags)
{
    __shared__ __attribute__((aligned(16))) float lds_all[4][wfx::kLdsFloats];
    wfx::wave_block<CH, CHN>(ctx, slot, slot_count, list, count, seg, flags, static_cast<int>(blockIdx.x), &lds_all[0][0], wfx::kLdsFloats);
    if constexpr (CH == 1) OALSFX_EQUAL_PLACES(); // (the stereo build takes 128 registers as it is, and spilled with the statement)
}

// ctx.turn != nullptr: a launch of a run of chained launches (mono / stereo): the build whose wavefronts take turns.
void launch_wave_effects(const KernelCtx& ctx, int slot, int slot_count, const int* list, int count, const WaveSegments* seg, int flags,
                         hipStream_t stream)
{
    if (count <= 0 || slot_count <= 0) return;
    WaveSegments s{};
    if (seg && slot_count == 1) s = *seg;
    const dim3 grid(s.n > 0 ? s.blocks() : (count + 3) / 4), block(256);
    if (ctx.turn != nullptr && ctx.channels <= 2) {
        if (ctx.channels == 1) OALSFX_LAUNCH((k_wave_effects<1, true>), grid, block, stream, ctx, slot, slot_count, list, count, s, flags);
        else OALSFX_LAUNCH((k_wave_effects<2, true>), grid, block, stream, ctx, slot, slot_count, list, count, s, flags);
        return;
    }
    if (ctx.channels == 1) OALSFX_LAUNCH((k_wave_effects<1, false>), grid, block, stream, ctx, slot, slot_count, list, count, s, flags);
    else if (ctx.channels == 2) OALSFX_LAUNCH((k_wave_effects<2, false>), grid, block, stream, ctx, slot, slot_count, list, count, s, flags);
    else OALSFX_LAUNCH((k_wave_effects<8, false>), grid, block, stream, ctx, slot, slot_count, list, count, s, flags);
}

} // namespace oalsfx_hip
