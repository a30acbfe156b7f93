"""CPU: host-side logic of the product -- property defaults / clamping / presets, the deferred/active apply rules
and the channel-format helpers -- against data recorded from the reference (tests/golden/effect_defaults.npz) and
against the reference's documented rules."""
import ctypes as C
import os

import numpy as np

from harness import OracleApi, make_effect
from oalsfxpp_amd import desc, lib

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_effect_defaults_match_the_reference():
    gold = np.load(os.path.join(GOLD, "effect_defaults.npz"))
    for t in range(12):
        e = lib.effect_defaults(t)
        raw = gold[desc.EFFECT_NAMES[t]].tobytes()
        assert e.type == t
        if t in desc.PROPS_MEMBER:
            m = desc.PROPS_MEMBER[t]
            g = desc.Effect.from_buffer_copy(raw)
            assert bytes(getattr(e.props, m)) == bytes(getattr(g.props, m)), desc.EFFECT_NAMES[t]


def test_normalize_clamps_every_field():
    e = make_effect(desc.EAX_REVERB, density=5.0, gain=-1.0, decay_time=1000.0, reflections_pan=[3, -3, 0], hf_reference=1.0)
    n = lib.effect_normalized(e).props.reverb
    assert (n.density, n.gain, n.decay_time, n.hf_reference) == (1.0, 0.0, 20.0, 1000.0)
    assert list(n.reflections_pan) == [1.0, -1.0, 0.0]
    c = lib.effect_normalized(make_effect(desc.CHORUS, waveform=7, phase=-999, rate=99.0)).props.chorus
    assert (c.waveform, c.phase, c.rate) == (1, -180, 10.0)


def test_presets():
    assert lib.preset_count() == 113
    name, p = lib.preset(0)
    assert name == "Default::generic" and abs(p.decay_time - 1.49) < 1e-6 and p.decay_hf_limit
    name, p = lib.preset(112)
    assert name == "Misc::small_water_room"
    names = {lib.preset(i)[0] for i in range(113)}
    assert len(names) == 113


def test_channel_counts():
    so = lib.load()
    assert [so.oalsfx_host_channel_count(f) for f in range(8)] == [0, 1, 2, 4, 6, 6, 7, 8]


def test_ring_sizes_at_48k():
    so = lib.load()
    # reverb 58,880 frames x 4 lines, chorus 2 x 2048, flanger 2 x 512, echo 32768
    assert so.oalsfx_host_ring_floats(desc.EAX_REVERB, 48000) == 58880 * 4
    assert so.oalsfx_host_ring_floats(desc.CHORUS, 48000) == 2 * 2048
    assert so.oalsfx_host_ring_floats(desc.FLANGER, 48000) == 2 * 512
    assert so.oalsfx_host_ring_floats(desc.ECHO, 48000) == 32768
    assert so.oalsfx_host_ring_floats(desc.EQUALIZER, 48000) == 0


def test_default_reverb_taps_at_48k():
    # SURVEY 8a rows a21/a22 [probe]: default tap positions of the reference at 48 kHz
    p = lib.derive_slot(desc.FMT_STEREO, 48000, lib.effect_normalized(lib.effect_defaults(desc.EAX_REVERB))).u.reverb
    assert list(p.early_tap) == [336, 821, 1356, 1948]
    assert list(p.early_ap_off) == [233, 257, 284, 313]
    assert list(p.early_line_off) == [1436, 2619, 3690, 4660]
    assert p.late_feed_tap == 16012
    assert list(p.late_tap) == [16540, 16842, 17402, 17705]
    assert list(p.late_ap_off) == [388, 489, 675, 776]
    assert list(p.late_line_off) == [4660, 5872, 8109, 9321]
    assert list(p.ring_len) == [32768, 512, 8192, 1024, 16384]


def test_stereo_dry_matrix_is_panned_not_identity():
    # SURVEY 8a row a3 [probe]: stereo L->L 0.92936, L->R 0.42936; aux L: [1, 0.8660, 0, 1.5]
    sp = lib.derive_source(desc.FMT_STEREO, 48000, desc.SendProps(1, 1, 1), [desc.SendProps(1, 1, 1)], [desc.EAX_REVERB])
    assert abs(sp.direct.gains[0][0] - 0.92936) < 1e-4 and abs(sp.direct.gains[0][1] - 0.42936) < 1e-4
    assert [round(x, 4) for x in sp.aux[0].gains[0][:4]] == [1.0, 0.866, 0.0, 1.5]
    assert sp.direct.filter_type == 0 and sp.aux[0].out_channels == 4


def test_null_slot_disables_its_send_and_lfe_quirks():
    sp = lib.derive_source(desc.FMT_5POINT1, 48000, desc.SendProps(1, 1, 1), [desc.SendProps(1, 1, 1)], [desc.NULL])
    assert sp.aux[0].out_channels == 0
    assert all(g == 0.0 for g in sp.direct.gains[3])          # LFE input channel: no dry gain (get_channel_index quirk)
    rear = lib.derive_source(desc.FMT_5POINT1_REAR, 48000, desc.SendProps(1, 1, 1), [desc.SendProps(1, 1, 1)], [desc.ECHO])
    assert all(g == 0.0 for row in rear.direct.gains for g in row)  # 5.1-rear is missing from the reference's switch
    ded = lib.derive_slot(desc.FMT_5POINT1, 48000, lib.effect_defaults(desc.DEDICATED_LFE)).u.dedicated
    assert all(g == 0.0 for g in ded.gains)                   # dedicated LFE is silent in the reference


def test_apply_rules_deferred_vs_active():
    api = OracleApi(desc.FMT_STEREO, 48000, 2)
    api.set_effect_type(0, desc.ECHO)
    assert api.active[0].type == desc.NULL                    # nothing happens before apply_changes
    api.apply_changes()
    assert api.active[0].type == desc.ECHO and api.slot_restart[0]
    api.refresh()
    api.set_effect(0, make_effect(desc.ECHO, delay=0.05))
    api.apply_changes()
    assert api.slot_changed[0] and not api.slot_restart[0]    # same type: properties only, state kept
    api.refresh()
    api.apply_changes()
    assert not api.slot_changed[0]                            # unchanged properties do not trigger an update
    api.set_send_props(1, 0.5, 1.0, 1.0)                      # aux send: active props written directly, re-fires every apply
    api.apply_changes()
    assert api.source_changed
    api.refresh()
    api.apply_changes()
    assert api.source_changed
