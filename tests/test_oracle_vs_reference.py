"""CPU, build container only: the oracle against the *live* compiled reference (oracle/_ref/libref.so) on cases
that are too many or too large to keep as fixtures: all 113 presets, every channel format, low and high
sampling rates, randomised properties.  Skipped where oracle/_ref is absent (e.g. on the GPU box)."""
import random

import numpy as np
import pytest

from harness import OracleApi, make_effect, preset_effect, same_bits, struct_diff
from oalsfxpp_amd import desc, lib
from oracle import oracle as orc

pytestmark = pytest.mark.skipif(not orc.have_reference(), reason="oracle/_ref not built (needs /root/reference)")


def compare(fmt, rate, slots, effects, mixes, sends=()):
    ref, mine = orc.Reference(fmt, rate, slots), OracleApi(fmt, rate, slots)
    for s, e in effects:
        ref.set_effect(s, e)
        mine.set_effect(s, e)
    for args in sends:
        ref.set_send_props(*args)
        mine.set_send_props(*args)
    ref.apply_changes()
    mine.apply_changes()
    for k, frames in enumerate(mixes):
        x = orc.synth(5, k, frames * ref.channels).reshape(frames, ref.channels)
        ok, nbad = same_bits(ref.mix(x), mine.mix(x))
        assert ok, f"mix {k}: {nbad} samples differ"
    for s in range(slots):
        rp, rs = ref.dump_slot(s)
        if rp.type in desc.PARAMS_MEMBER:
            m = desc.PARAMS_MEMBER[rp.type]
            assert not struct_diff(getattr(rp.u, m), getattr(mine.params[s].u, m), m)
        if rp.type in desc.STATE_MEMBER:
            m = desc.STATE_MEMBER[rp.type]
            assert not struct_diff(getattr(rs.u, m), getattr(mine.oracle.state(s).u, m), m)
        ok, nbad = same_bits(ref.dump_rings(s, rp), mine.oracle.ring(s))
        assert ok, f"slot {s}: {nbad} ring words differ"
    sp, ss = ref.dump_source()
    assert not struct_diff(sp, mine.source_params, "source")
    assert not struct_diff(ss, mine.oracle.source_state(), "source_state")


@pytest.mark.parametrize("index", range(0, 113))
def test_every_preset(index):
    compare(desc.FMT_STEREO, 48000, 1, [(0, preset_effect(index))], [256] * 3)


@pytest.mark.parametrize("fmt", range(1, 8))
def test_every_format_every_effect(fmt):
    for t in range(12):
        compare(fmt, 44100, 1, [(0, make_effect(t))], [200, 56])


@pytest.mark.parametrize("rate", [8000, 11025, 22050, 32000, 96000, 192000])
def test_rates(rate):
    for t in (desc.EAX_REVERB, desc.REVERB, desc.CHORUS, desc.FLANGER, desc.ECHO, desc.DISTORTION, desc.RING_MODULATOR, desc.EQUALIZER, desc.COMPRESSOR):
        compare(desc.FMT_STEREO, rate, 1, [(0, make_effect(t))], [256] * 3)


from oalsfxpp_amd.workloads import FIELDS, random_effect  # noqa: E402


@pytest.mark.parametrize("seed", range(24))
def test_random_properties(seed):
    rng = random.Random(seed)
    t = rng.choice(list(FIELDS))
    compare(desc.FMT_STEREO, 48000, 1, [(0, random_effect(rng, t))], [256] * 5)
    compare(desc.FMT_MONO, 48000, 1, [(0, random_effect(rng, desc.EAX_REVERB))], [256] * 4)
