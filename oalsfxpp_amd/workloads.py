"""Synthetic workloads of BASELINE.json / SURVEY 8d, built through the public property API only.

Property ranges are the limits of the reference header (src/oalsfxpp.h:76-503); `random_effect` draws every field
uniformly inside them (config 4's "randomised params").
"""
import random

from . import desc, lib

_REVERB_FIELDS = dict(
    density=(0.0, 1.0), diffusion=(0.0, 1.0), gain=(0.0, 1.0), gain_hf=(0.0, 1.0), gain_lf=(0.0, 1.0), decay_time=(0.1, 20.0),
    decay_hf_ratio=(0.1, 2.0), decay_lf_ratio=(0.1, 2.0), reflections_gain=(0.0, 3.16), reflections_delay=(0.0, 0.3),
    late_reverb_gain=(0.0, 10.0), late_reverb_delay=(0.0, 0.1), echo_time=(0.075, 0.25), echo_depth=(0.0, 1.0),
    modulation_time=(0.04, 4.0), modulation_depth=(0.0, 1.0), air_absorption_gain_hf=(0.892, 1.0), hf_reference=(1000.0, 20000.0),
    lf_reference=(20.0, 1000.0))

FIELDS = {
    desc.CHORUS: dict(waveform=(0, 1), phase=(-180, 180), rate=(0.0, 10.0), depth=(0.0, 1.0), feedback=(-1.0, 1.0), delay=(0.0, 0.016)),
    desc.FLANGER: dict(waveform=(0, 1), phase=(-180, 180), rate=(0.0, 10.0), depth=(0.0, 1.0), feedback=(-1.0, 1.0), delay=(0.0, 0.004)),
    desc.COMPRESSOR: dict(on_off=(0, 1)),
    desc.DEDICATED_DIALOG: dict(gain=(0.0, 1.0)),
    desc.DEDICATED_LFE: dict(gain=(0.0, 1.0)),
    desc.DISTORTION: dict(edge=(0.0, 1.0), gain=(0.01, 1.0), low_pass_cutoff=(80.0, 24000.0), eq_center=(80.0, 24000.0), eq_bandwidth=(80.0, 24000.0)),
    desc.ECHO: dict(delay=(0.0, 0.207), lr_delay=(0.0, 0.404), damping=(0.0, 0.99), feedback=(0.0, 1.0), spread=(-1.0, 1.0)),
    desc.EQUALIZER: dict(low_gain=(0.126, 7.943), low_cutoff=(50.0, 800.0), mid1_gain=(0.126, 7.943), mid1_center=(200.0, 3000.0), mid1_width=(0.01, 1.0),
                         mid2_gain=(0.126, 7.943), mid2_center=(1000.0, 8000.0), mid2_width=(0.01, 1.0), high_gain=(0.126, 7.943), high_cutoff=(4000.0, 16000.0)),
    desc.RING_MODULATOR: dict(frequency=(0.0, 8000.0), high_pass_cutoff=(0.0, 24000.0), waveform=(0, 2)),
    desc.REVERB: _REVERB_FIELDS,
    desc.EAX_REVERB: _REVERB_FIELDS,
}


def make_effect(effect_type, **fields):
    """Default properties of `effect_type` with some fields overridden (un-normalised)."""
    e = lib.effect_defaults(effect_type)
    if fields:
        member = getattr(e.props, desc.PROPS_MEMBER[effect_type])
        for k, v in fields.items():
            if isinstance(v, (list, tuple)):
                arr = getattr(member, k)
                for i, x in enumerate(v):
                    arr[i] = x
            else:
                setattr(member, k, v)
    return e


def random_effect(rng, t):
    """Every field uniform in its [min, max]; integer fields uniform over their range."""
    over = {}
    for k, (lo, hi) in FIELDS.get(t, {}).items():
        over[k] = rng.randint(lo, hi) if isinstance(lo, int) else rng.uniform(lo, hi)
    if t in (desc.EAX_REVERB, desc.REVERB):
        over["reflections_pan"] = [rng.uniform(-1, 1) for _ in range(3)]
        over["late_reverb_pan"] = [rng.uniform(-1, 1) for _ in range(3)]
        over["decay_hf_limit"] = rng.random() < 0.5
    return make_effect(t, **over)


# ---- BASELINE.json configs as (effect_count, setup(batch, first_global_instance)) ----
CONFIG3_CHAIN = (desc.CHORUS, desc.FLANGER, desc.ECHO, desc.EAX_REVERB)


def config4_type(instance):
    """type = 1 + instance % 11: every non-null effect type."""
    return 1 + instance % 11


def setup(batch, name, first_instance=0):
    """Programs `batch` for a named workload.  `first_instance` is this rank's global offset (seeds stay global)."""
    n = batch.n
    if name == "config2":
        batch.set_effect_type(0, desc.EAX_REVERB)
    elif name == "config2-presets":
        effects = []
        for i in range(n):
            e = lib.effect_defaults(desc.EAX_REVERB)
            e.props.reverb = lib.preset((first_instance + i) % lib.preset_count())[1]
            effects.append(e)
        batch.set_effect(0, effects)
    elif name == "config3":
        for slot, t in enumerate(CONFIG3_CHAIN):
            batch.set_effect_type(slot, t)
    elif name == "config4":
        batch.set_effect(0, [random_effect(random.Random(first_instance + i), config4_type(first_instance + i)) for i in range(n)])
    else:
        raise ValueError(name)
    batch.apply_changes()


def effect_count(name):
    return 4 if name == "config3" else 1


# SURVEY 8d algorithmic bytes per stereo frame (16 B I/O included once per instance)
BYTES_PER_FRAME = {
    desc.NULL: 16, desc.CHORUS: 32, desc.FLANGER: 32, desc.ECHO: 28, desc.EQUALIZER: 16, desc.DISTORTION: 16,
    desc.RING_MODULATOR: 16, desc.COMPRESSOR: 16, desc.DEDICATED_DIALOG: 16, desc.DEDICATED_LFE: 16,
    desc.REVERB: 208, desc.EAX_REVERB: 208,
}
CONFIG3_BYTES_PER_FRAME = 252   # 16 + 16 + 16 + 12 + 192
