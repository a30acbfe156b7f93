"""GPU parity of the send shelf filters inside the steady-state reverb kernel (build flag SF): single-slot batches whose
reverbs are proven steady filter their own sends (reference apply_filters, src/oalsfxpp.cpp:3101-3143; pass-through histories
:1038-1056) in a tile loop that is skewed once more, instead of reading the planes of the pre-pass kernel.

What has to hold: outputs, effect state, delay lines and the send filters' histories bit-identical to the oracle whichever path
filtered an instance; the SF builds are the ones that run once the instances are proven; instances of the other kinds (short taps,
modulated, believed, cross-fading) and calls the SF builds do not take (ragged ones) still come out right beside them."""
import numpy as np
import pytest

from harness import make_effect, preset_effect, steady_build
from oalsfxpp_amd import desc
from test_gpu_proven import Follow

pytestmark = pytest.mark.gpu

E = make_effect


def sf_build(symbol):
    """Does the symbol name a build with the send filters inside?  (k_reverb_steady_coop: 11th template argument; the grid of kinds: 3rd.)"""
    args = [a.strip() for a in symbol[symbol.index("<") + 1: symbol.rindex(">")].split(",")]
    return args[2] == "true" if symbol.startswith("k_reverb_steady_kinds") else (len(args) > 10 and args[10] == "true")


SENDS = [(-1, 0.9, 0.5, 1.0), (0, 0.8, 1.0, 0.4), (-1, 1.0, 0.3, 0.6), (0, 0.7, 0.25, 0.5), (-1, 0.6, 1.0, 0.2)]


@pytest.mark.parametrize("fmt", [desc.FMT_MONO, desc.FMT_STEREO])
@pytest.mark.parametrize("presets", [[0], [2, 0], [0, 4, 2, 12, 3, 23, 26, 60]])
def test_filters_inside_the_steady_state_builds(fmt, presets):
    """Plain presets alone, plain and close-tap ones, and a mix with short-tap and modulated ones (the grid of kinds: the first two kinds
    filter inside, the others read the pre-pass planes), some instances without any filter, every combination of shelves on the
    direct and the auxiliary send."""
    n = 16
    f = Follow(fmt, 48000, 1, [[(0, preset_effect(presets[i % len(presets)], desc.EAX_REVERB if i % 3 else desc.REVERB))] for i in range(n)])
    try:
        b = f.b
        for i in range(n):
            if i % 4 == 3:
                continue                      # no filter at all
            for k in range(1 + i % 2):
                slot, g, hf, lf = SENDS[(i + k) % len(SENDS)]
                b.set_send_props(slot, g, hf, lf, first=i, count=1)
        f.apply()
        for frames in (256, 256, 256):
            f.mix(frames)
        assert b.plan(0)[1] == n, b.plan(0)
        for frames in (256, 256, 64, 128, 2048, 256):
            f.mix(frames)
            assert sf_build(b.last_reverb_kernel), b.last_reverb_kernel
        f.check_state()
        f.mix(100)                            # a ragged call: the pre-pass for everyone
        f.mix(256); f.mix(256); f.mix(256)
        assert sf_build(b.last_reverb_kernel), b.last_reverb_kernel
        # filters switched off, others switched on, gains changed: the histories must carry over either way
        b.set_send_props(-1, 1.0, 1.0, 1.0, first=0, count=3)
        b.set_send_props(0, 1.0, 1.0, 1.0, first=1, count=2)
        b.set_send_props(-1, 0.5, 0.9, 0.1, first=3, count=1)
        b.set_send_props(0, 0.9, 0.2, 1.0, first=7, count=1)
        f.apply()
        for frames in (256, 256, 256, 64, 256):
            f.mix(frames)
        f.check_state()
        # a property change among them: that instance cross-fades on the XF build beside the SF ones
        b.set_effect(0, preset_effect(40), first=5, count=1)
        f.apply()
        for frames in (256, 256, 256, 256):
            f.mix(frames)
        f.check_state()
    finally:
        f.close()


def test_every_instance_filtered_leaves_no_pre_pass_and_one_filtered_instance_among_many():
    """What scripts/send_filter_bench.py times, followed by the oracle."""
    n = 64
    f = Follow(desc.FMT_STEREO, 48000, 1, [[(0, E(desc.EAX_REVERB))] for _ in range(n)])
    try:
        b = f.b
        for _ in range(3):
            f.mix(256)
        b.set_send_props(-1, 1.0, 0.5, 1.0)
        f.apply()
        for _ in range(4):
            f.mix(256)
        assert sf_build(b.last_reverb_kernel) and steady_build(b.last_reverb_kernel)["fp"], b.last_reverb_kernel
        b.set_send_props(-1, 1.0, 1.0, 1.0)
        b.set_send_props(-1, 0.8, 0.4, 0.7, first=17, count=1)
        b.set_send_props(0, 0.8, 0.4, 0.7, first=17, count=1)
        f.apply()
        for _ in range(4):
            f.mix(256)
        assert sf_build(b.last_reverb_kernel), b.last_reverb_kernel
        f.check_state()
    finally:
        f.close()
