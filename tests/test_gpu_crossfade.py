"""GPU parity of property changes while streaming on the cooperative reverb kernel (build flag XF): a settled reverb whose
properties change cross-fades its taps over 128 frames (reference src/oalsfxpp.cpp:6062-6075, 6088-6096, 6118-6138, 7378-7399) and
ramps its output gains (src/oalsfxpp.cpp:2752-2798) inside its 4-wave workgroup instead of on the one-wavefront general kernel.

What has to hold: outputs, effect state and delay lines bit-identical to the oracle through every kind of change and call size;
changes the build can follow never reach the general kernel (the launch plan says so); changes it cannot follow (taps shorter
than a tile in either set, a type change, a fade that does not stand at a tile boundary) still come out right."""
import numpy as np
import pytest

from harness import crossfade_followable, make_effect, preset_effect, reverb_params, steady_build
from oalsfxpp_amd import desc, lib
from test_gpu_proven import Follow

pytestmark = pytest.mark.gpu

E = make_effect


def followable_pairs(limit, start=0):
    """(from, to) preset pairs the XF build can follow, and some it cannot."""
    params = [reverb_params(preset_effect(i)) for i in range(113)]
    yes, no = [], []
    for a in range(start, 113):
        for b in (7 * a + 3) % 113, (11 * a + 50) % 113:
            if a != b:
                (yes if crossfade_followable(params[a], params[b]) else no).append((a, b))
    return yes[:limit], no[:limit]


@pytest.mark.parametrize("fmt", [desc.FMT_MONO, desc.FMT_STEREO])
@pytest.mark.parametrize("frames", [256, 64, 512, 128, 2048])
def test_preset_changes_stay_on_the_cooperative_kernel(fmt, frames):
    yes, no = followable_pairs(14)
    pairs = yes + no[:4]
    f = Follow(fmt, 48000, 1, [[(0, preset_effect(a))] for a, _ in pairs])
    try:
        b = f.b
        for _ in range(3):
            f.mix(256)
        assert b.plan(0)[3] == 0
        for i, (_, to) in enumerate(pairs):
            b.set_effect(0, preset_effect(to), first=i, count=1)
        f.apply()
        plan = b.plan(0)
        assert plan[3] == len(no[:4]), f"{plan}: the changes the XF build can follow are listed for the steady-state kernel"
        f.mix(frames)
        assert steady_build(b.last_reverb_kernel)["xf"], b.last_reverb_kernel
        for _ in range(max(2, 512 // frames)):
            f.mix(frames)
        f.mix(256); f.mix(256)
        assert b.plan(0)[3] == 0 and b.plan(0)[1] == len(pairs), b.plan(0)   # all settled and proven again
        f.check_state()
    finally:
        f.close()


def test_gain_only_changes_ramp_without_a_fade():
    """Properties that move output gains and filter coefficients but no tap: no cross-fade, gain ramps over the call
    (MixHelpers::mix, reference src/oalsfxpp.cpp:2752-2798), coefficients at once."""
    n = 6
    f = Follow(desc.FMT_STEREO, 48000, 1, [[(0, E(desc.EAX_REVERB))] for _ in range(n)])
    try:
        b = f.b
        for _ in range(3):
            f.mix(256)
        b.set_effect(0, E(desc.EAX_REVERB, gain=0.1), first=0, count=1)
        b.set_effect(0, E(desc.EAX_REVERB, reflections_pan=[0.5, 0.0, -0.5], late_reverb_pan=[-0.3, 0.0, 0.6]), first=1, count=1)
        b.set_effect(0, E(desc.EAX_REVERB, gain_hf=0.2, gain_lf=0.5, decay_time=6.0), first=2, count=1)
        b.set_effect(0, E(desc.EAX_REVERB, late_reverb_gain=4.0, reflections_gain=0.0), first=3, count=1)
        f.apply()
        assert b.plan(0)[3] == 0
        f.mix(512)     # two blocks: the ramps' steps are taken anew at the second
        f.mix(256)
        b.set_effect(0, E(desc.EAX_REVERB, gain=1.0), first=0, count=1)
        f.apply()
        f.mix(64); f.mix(64); f.mix(192); f.mix(256)
        assert b.plan(0)[3] == 0
        f.check_state()
    finally:
        f.close()


def test_changes_in_quick_succession_and_ragged_calls_in_between():
    """A second change before the first fade is over restarts the fade from the taps that are still current (reference
    src/oalsfxpp.cpp:6062-6075); a ragged call in the middle of a fade leaves the fade count off the tile grid, which only the
    general path handles; changing back and forth between two presets."""
    f = Follow(desc.FMT_STEREO, 48000, 1, [[(0, preset_effect(p))] for p in (0, 4, 12, 30, 60, 90)])
    try:
        b = f.b
        for _ in range(3):
            f.mix(256)
        for i, to in enumerate((12, 30, 60, 90, 0, 4)):
            b.set_effect(0, preset_effect(to), first=i, count=1)
        f.apply()
        f.mix(64)                                    # half of the fade
        b.set_effect(0, preset_effect(17), first=0, count=2)
        f.apply()                                    # a new target while fading: the fade restarts
        f.mix(64); f.mix(64); f.mix(256)
        b.set_effect(0, preset_effect(5), first=2, count=2)
        f.apply()
        f.mix(100)                                   # ragged: fade count 100
        f.mix(256); f.mix(256)
        for k in range(6):
            b.set_effect(0, preset_effect(40 if k % 2 == 0 else 41), first=4, count=2)
            f.apply()
            f.mix(256)
        f.mix(256); f.mix(256)
        assert b.plan(0)[3] == 0
        f.check_state()
    finally:
        f.close()


def test_changes_in_a_slot_shared_with_ring_light_effects_and_in_a_later_slot():
    """The grid that serves a slot's ring-light effects and steady reverbs together (k_slot_mixed) follows changes too; and a reverb in
    a later slot (running mix through mixbuf), with a send filter on one instance."""
    setups = []
    for i in range(12):
        first = E(desc.CHORUS) if i % 3 == 0 else preset_effect(4 + i)
        setups.append([(0, first), (1, preset_effect(20 + i))])
    f = Follow(desc.FMT_STEREO, 48000, 2, setups)
    try:
        b = f.b
        b.set_send_props(-1, 0.9, 0.5, 1.0, first=1, count=1)
        b.set_send_props(1, 0.8, 1.0, 0.4, first=2, count=1)
        f.apply()
        for _ in range(3):
            f.mix(256)
        for i in range(12):
            if i % 3 != 0:
                b.set_effect(0, preset_effect(60 + i), first=i, count=1)
            if i % 2 == 0:
                b.set_effect(1, preset_effect(80 + i), first=i, count=1)
        f.apply()
        f.mix(256); f.mix(256); f.mix(256); f.mix(256)
        f.check_state()
    finally:
        f.close()


def test_a_storm_of_changes_among_many_instances():
    """What scripts/update_storm_bench.py times: a few of many reverbs get a new preset before every buffer.  Every instance is
    followed; no call may leave more than the unfollowable changes to the general kernel."""
    import random
    n = 192
    rng = random.Random(5)
    f = Follow(desc.FMT_STEREO, 48000, 1, [[(0, preset_effect(i % 113))] for i in range(n)])
    try:
        b = f.b
        params = [reverb_params(preset_effect(i)) for i in range(113)]
        current = [i % 113 for i in range(n)]
        settled_at = [0] * n
        for _ in range(3):
            f.mix(256)
        for step in range(10):
            expect_general = 0
            for _ in range(6):
                i, to = rng.randrange(n), rng.randrange(113)
                if to == current[i]:
                    continue
                # (an instance changed again before it settled goes by the general kernel: not counted here)
                ok = crossfade_followable(params[current[i]], params[to]) and settled_at[i] <= step
                expect_general += 0 if ok else 1
                b.set_effect(0, preset_effect(to), first=i, count=1)
                current[i], settled_at[i] = to, step + 1
            f.apply()
            assert b.plan(0)[3] <= expect_general, (step, b.plan(0), expect_general)
            f.mix(256)
        f.mix(256); f.mix(256)
        assert b.plan(0)[3] == 0
        f.check_state()
    finally:
        f.close()


def test_set_effect_at_is_set_effect_per_instance():
    """`oalsfx_batch_set_effect_at` (instances that are not neighbours, one foreign call) against one `set_effect` call per instance:
    same descriptors, same outputs; an index out of range sets nothing and reports."""
    import pytest as _pytest
    from oalsfxpp_amd.api import Batch, BatchError
    n = 16
    picks, to = [14, 3, 9], [20, 57, 101]
    x = np.stack([np.sin(np.arange(256 * 2, dtype=np.float32) * (0.01 + 0.001 * i)).reshape(256, 2) for i in range(n)]).astype(np.float32)
    outs = []
    for at in (False, True):
        with Batch(n, desc.FMT_STEREO, 48000, 1) as b:
            b.set_effect(0, [preset_effect(i) for i in range(n)])
            b.apply_changes()
            for _ in range(3):
                b.mix(x)
            if at:
                with _pytest.raises(BatchError):
                    b.set_effect_at(0, [2, n], [preset_effect(1), preset_effect(2)])
                b.set_effect_at(0, picks, [preset_effect(p) for p in to])
            else:
                for i, p in zip(picks, to):
                    b.set_effect(0, preset_effect(p), first=i, count=1)
            b.apply_changes()
            ys = [b.mix(x).copy() for _ in range(3)]
            params = [bytes(b.read_slot(i, 0)[0])[8:] for i in range(n)]   # (behind type and update_seq)
            outs.append((ys, params))
    for ya, yb in zip(outs[0][0], outs[1][0]):
        assert (ya.view(np.uint32) == yb.view(np.uint32)).all()
    assert outs[0][1] == outs[1][1]
