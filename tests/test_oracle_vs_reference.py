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


FIELDS = {
    desc.CHORUS: dict(waveform=(0, 1), phase=(-180, 180), rate=(0.0, 10.0), depth=(0.0, 1.0), feedback=(-1.0, 1.0), delay=(0.0, 0.016)),
    desc.FLANGER: dict(waveform=(0, 1), phase=(-180, 180), rate=(0.0, 10.0), depth=(0.0, 1.0), feedback=(-1.0, 1.0), delay=(0.0, 0.004)),
    desc.DISTORTION: dict(edge=(0.0, 1.0), gain=(0.01, 1.0), low_pass_cutoff=(80.0, 24000.0), eq_center=(80.0, 24000.0), eq_bandwidth=(80.0, 24000.0)),
    desc.ECHO: dict(delay=(0.0, 0.207), lr_delay=(0.0, 0.404), damping=(0.0, 0.99), feedback=(0.0, 1.0), spread=(-1.0, 1.0)),
    desc.EQUALIZER: dict(low_gain=(0.126, 7.943), low_cutoff=(50.0, 800.0), mid1_gain=(0.126, 7.943), mid1_center=(200.0, 3000.0), mid1_width=(0.01, 1.0),
                         mid2_gain=(0.126, 7.943), mid2_center=(1000.0, 8000.0), mid2_width=(0.01, 1.0), high_gain=(0.126, 7.943), high_cutoff=(4000.0, 16000.0)),
    desc.RING_MODULATOR: dict(frequency=(0.0, 8000.0), high_pass_cutoff=(0.0, 24000.0), waveform=(0, 2)),
    desc.EAX_REVERB: dict(density=(0.0, 1.0), diffusion=(0.0, 1.0), gain=(0.0, 1.0), gain_hf=(0.0, 1.0), gain_lf=(0.0, 1.0), decay_time=(0.1, 20.0),
                          decay_hf_ratio=(0.1, 2.0), decay_lf_ratio=(0.1, 2.0), reflections_gain=(0.0, 3.16), reflections_delay=(0.0, 0.3),
                          late_reverb_gain=(0.0, 10.0), late_reverb_delay=(0.0, 0.1), echo_time=(0.075, 0.25), echo_depth=(0.0, 1.0),
                          modulation_time=(0.04, 4.0), modulation_depth=(0.0, 1.0), air_absorption_gain_hf=(0.892, 1.0), hf_reference=(1000.0, 20000.0),
                          lf_reference=(20.0, 1000.0)),
}


def random_effect(rng, t):
    """Every field uniform in its [min, max] (BASELINE config 4's parameter randomisation)."""
    over = {}
    for k, (lo, hi) in FIELDS.get(t, {}).items():
        over[k] = rng.randint(lo, hi) if isinstance(lo, int) else rng.uniform(lo, hi)
    if t in (desc.EAX_REVERB, desc.REVERB):
        over["reflections_pan"] = [rng.uniform(-1, 1) for _ in range(3)]
        over["late_reverb_pan"] = [rng.uniform(-1, 1) for _ in range(3)]
        over["decay_hf_limit"] = rng.random() < 0.5
    return make_effect(t, **over)


@pytest.mark.parametrize("seed", range(24))
def test_random_properties(seed):
    rng = random.Random(seed)
    t = rng.choice(list(FIELDS))
    compare(desc.FMT_STEREO, 48000, 1, [(0, random_effect(rng, t))], [256] * 5)
    compare(desc.FMT_MONO, 48000, 1, [(0, random_effect(rng, desc.EAX_REVERB))], [256] * 4)
