#!/usr/bin/env python3
"""Generates the golden vectors in tests/golden/ from the compiled reference (oracle/_ref/libref.so).

Run in the build container only (it needs /root/reference to build oracle/_ref):

    make -C oracle ref && python tests/golden/generate.py

What is stored (data only -- no reference source):
  * cases.json          the call script of every case (effect type, property overrides, frame counts)
  * <case>.npz          `out`  : concatenated interleaved float32 output of all mix calls
                        `params<slot>` / `source`: the reference's derived parameters after the last
                        refresh, in this repository's descriptor layout (raw bytes), so the host update
                        path can be checked without the reference
                        `state<slot>`: final effect state; `ring<slot>`: CRC32 of the final delay rings
  * effect_defaults.npz the 112-byte Effect of every type after set_type_and_defaults
  * sinf_bits.npz       sinf() of the container's libm on a fixed argument grid (pins oracle/ref_sinf.h)

Inputs are not stored: they come from the repository's integer PRNG (oracle_synth) seeded by
(case seed, mix index).
"""
import ctypes as C
import json
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from harness import make_effect, preset_effect  # noqa: E402
from oalsfxpp_amd import desc  # noqa: E402
from oracle import oracle as orc  # noqa: E402

D = desc


def cases():
    """name -> dict(fmt, rate, slots, seed, script).  Script ops are JSON-able:
    ["effect", slot, type, {overrides}] / ["preset", slot, type, index] / ["send", slot, g, ghf, glf] / ["apply"] / ["mix", frames]"""
    out = {}
    mix8 = [["mix", 256]] * 8
    for t in range(12):
        for fmt, tag in ((D.FMT_MONO, "mono"), (D.FMT_STEREO, "stereo")):
            out[f"default_{D.EFFECT_NAMES[t]}_{tag}"] = dict(fmt=fmt, rate=48000, slots=1, seed=100 + t,
                                                            script=[["effect", 0, t, {}], ["apply"]] + mix8)
    S, R = D.FMT_STEREO, 48000
    def one(name, op, mixes=None, fmt=S, rate=R, seed=7):
        out[name] = dict(fmt=fmt, rate=rate, slots=1, seed=seed, script=[op, ["apply"]] + (mixes or [["mix", 256]] * 6))
    for i in (8, 25, 23, 112, 60, 95):
        one(f"eax_preset_{i}", ["preset", 0, D.EAX_REVERB, i])
    one("reverb_preset_25_mono", ["preset", 0, D.REVERB, 25], fmt=D.FMT_MONO)
    one("eax_modulated_dense", ["effect", 0, D.EAX_REVERB, dict(modulation_depth=1.0, modulation_time=0.04, echo_depth=0.7, echo_time=0.08,
        density=0.0, diffusion=0.3, reflections_delay=0.0, late_reverb_delay=0.0, gain_lf=0.3, reflections_pan=[0.3, -0.2, 0.5],
        late_reverb_pan=[-0.6, 0.1, -0.4], decay_hf_limit=False, decay_hf_ratio=2.0, decay_lf_ratio=0.3)], [["mix", 256]] * 10)
    one("eax_max_delays", ["effect", 0, D.EAX_REVERB, dict(reflections_delay=0.3, density=1.0, late_reverb_delay=0.1, modulation_depth=0.5,
        modulation_time=4.0, decay_time=20.0, decay_lf_ratio=2.0, decay_hf_ratio=0.1)], [["mix", 256]] * 8)
    out["eax_midstream_change_odd_sizes"] = dict(fmt=S, rate=R, slots=1, seed=11, script=[
        ["effect", 0, D.EAX_REVERB, {}], ["apply"]] + [["mix", 256]] * 4 + [["preset", 0, D.EAX_REVERB, 8], ["apply"]] + [["mix", 256]] * 3 +
        [["mix", 100], ["mix", 1], ["mix", 2], ["mix", 127], ["mix", 129], ["mix", 2500]])
    for w in (0, 1):
        for ph in (-180, 90):
            one(f"chorus_w{w}_ph{ph}", ["effect", 0, D.CHORUS, dict(waveform=w, phase=ph, rate=7.3, depth=0.9, feedback=-0.8, delay=0.011)])
            one(f"flanger_w{w}_ph{ph}", ["effect", 0, D.FLANGER, dict(waveform=w, phase=ph, rate=3.1, depth=1.0, feedback=0.9, delay=0.004)])
    for wv in (1, 2):
        one(f"ringmod_w{wv}", ["effect", 0, D.RING_MODULATOR, dict(waveform=wv, frequency=1234.5, high_pass_cutoff=3000.0)])
    one("compressor_off", ["effect", 0, D.COMPRESSOR, dict(on_off=False)])
    for sp in (-0.4, 1.0):
        one(f"echo_spread_{sp}", ["effect", 0, D.ECHO, dict(spread=sp, delay=0.01, lr_delay=0.02, damping=0.9, feedback=0.95)])
    for ed in (0.0, 1.0):
        one(f"distortion_edge_{ed}", ["effect", 0, D.DISTORTION, dict(edge=ed, gain=1.0, low_pass_cutoff=24000.0, eq_center=80.0, eq_bandwidth=24000.0)])
    one("equalizer_extreme", ["effect", 0, D.EQUALIZER, dict(low_gain=7.943, low_cutoff=50.0, mid1_gain=0.126, mid1_width=0.01, mid2_gain=7.0,
                                                            mid2_center=8000.0, high_gain=0.126, high_cutoff=16000.0)])
    one("eax_out_of_range_props", ["effect", 0, D.EAX_REVERB, dict(density=5.0, gain=-1.0, decay_time=1000.0, reflections_pan=[3, -3, 0])], [["mix", 256]] * 3)
    out["four_slots_config3"] = dict(fmt=S, rate=R, slots=4, seed=21, script=[
        ["effect", 0, D.CHORUS, {}], ["effect", 1, D.FLANGER, {}], ["effect", 2, D.ECHO, {}], ["effect", 3, D.EAX_REVERB, {}], ["apply"]] + mix8)
    out["send_filters_toggled"] = dict(fmt=S, rate=R, slots=2, seed=22, script=[
        ["effect", 0, D.EAX_REVERB, {}], ["effect", 1, D.ECHO, {}], ["send", -1, 0.7, 0.5, 1.0], ["send", 0, 0.9, 1.0, 0.3], ["send", 1, 0.5, 0.2, 0.4],
        ["apply"]] + [["mix", 256]] * 4 + [["send", -1, 1.0, 1.0, 1.0], ["send", 0, 1.0, 1.0, 1.0], ["apply"]] + [["mix", 256]] * 3)
    for fmt in (D.FMT_QUAD, D.FMT_5POINT1, D.FMT_5POINT1_REAR, D.FMT_6POINT1, D.FMT_7POINT1):
        for t in (D.EAX_REVERB, D.CHORUS, D.DEDICATED_DIALOG):
            one(f"fmt{fmt}_{D.EFFECT_NAMES[t]}", ["effect", 0, t, {}], [["mix", 200]] * 2, fmt=fmt, rate=44100)
    for rate in (11025, 96000):
        for t in (D.EAX_REVERB, D.FLANGER, D.ECHO):
            one(f"rate{rate}_{D.EFFECT_NAMES[t]}", ["effect", 0, t, {}], [["mix", 256]] * 4, rate=rate)
    out["type_changes"] = dict(fmt=S, rate=R, slots=1, seed=23, script=[
        ["effect", 0, D.ECHO, {}], ["apply"]] + [["mix", 256]] * 3 + [["effect", 0, D.EAX_REVERB, {}], ["apply"]] + [["mix", 256]] * 3 +
        [["effect", 0, D.ECHO, {}], ["apply"]] + [["mix", 256]] * 2 + [["effect", 0, D.NULL, {}], ["apply"], ["mix", 256]])
    return out


def build_effect(op):
    if op[0] == "preset":
        return preset_effect(op[3], op[2])
    return make_effect(op[2], **op[3])


def run_case(api, case, effect_setter):
    """Runs the script on an Api-like object (Reference or harness.OracleApi); returns concatenated outputs."""
    outs = []
    k = 0
    for op in case["script"]:
        if op[0] in ("effect", "preset"):
            effect_setter(api, op[1], build_effect(op))
        elif op[0] == "send":
            api.set_send_props(op[1], op[2], op[3], op[4])
        elif op[0] == "apply":
            api.apply_changes()
        elif op[0] == "mix":
            x = orc.synth(case["seed"], k, op[1] * desc.FORMAT_CHANNELS[case["fmt"]]).reshape(op[1], -1)
            outs.append(api.mix(x).reshape(-1))
            k += 1
    return np.concatenate(outs)


def main():
    all_cases = cases()
    json.dump(all_cases, open(os.path.join(HERE, "cases.json"), "w"), indent=0)
    total = 0
    for name, case in all_cases.items():
        ref = orc.Reference(case["fmt"], case["rate"], case["slots"])
        out = run_case(ref, case, lambda a, s, e: a.set_effect(s, e))
        ref.refresh()
        blobs = {"out": out}
        sp, _ = ref.dump_source()
        blobs["source"] = np.frombuffer(bytes(sp), dtype=np.uint8)
        for s in range(case["slots"]):
            p, st = ref.dump_slot(s)
            blobs[f"params{s}"] = np.frombuffer(bytes(p), dtype=np.uint8)
            blobs[f"state{s}"] = np.frombuffer(bytes(st), dtype=np.uint8)
            blobs[f"ring{s}"] = np.array([zlib.crc32(ref.dump_rings(s, p).tobytes())], dtype=np.uint32)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **blobs)
        total += os.path.getsize(path)
    defaults = {D.EFFECT_NAMES[t]: np.frombuffer(bytes(orc.ref_effect_defaults(t)), dtype=np.uint8) for t in range(12)}
    np.savez_compressed(os.path.join(HERE, "effect_defaults.npz"), **defaults)
    # libm sinf on a fixed grid: the argument ranges the process path produces ([-pi, 2 pi]) plus a coarse wide sweep
    libm = C.CDLL("libm.so.6")
    libm.sinf.restype = C.c_float
    libm.sinf.argtypes = [C.c_float]
    args = np.concatenate([np.linspace(-3.2, 6.4, 20001, dtype=np.float32), np.linspace(-100, 100, 4001, dtype=np.float32),
                           np.float32(2.0) ** np.arange(-20, 6, dtype=np.float32)])
    vals = np.array([libm.sinf(float(a)) for a in args], dtype=np.float32)
    np.savez_compressed(os.path.join(HERE, "sinf_bits.npz"), args=args, bits=vals.view(np.uint32))
    print(f"{len(all_cases)} cases, {total / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
