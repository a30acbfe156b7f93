"""Shared helpers for the parity tests.

`OracleApi` drives the CPU oracle (oracle/liboracle.so) through the same call sequence a caller
would use on `oalsfxpp::Api`: deferred setters, apply_changes, mix.  The parameter side goes
through the *product's* host update path (oalsfx_host_* entry points of liboalsfx_hip.so); the
sample side is the oracle.  It mirrors the reference's apply/refresh rules
(Api::apply_changes, reference src/oalsfxpp.cpp:3738-3783; update_context_sources, :3397-3412).
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oalsfxpp_amd import desc, lib  # noqa: E402
from oalsfxpp_amd.workloads import make_effect  # noqa: E402,F401
from oracle import oracle as orc  # noqa: E402


def preset_effect(index, effect_type=desc.EAX_REVERB):
    _, p = lib.preset(index)
    e = lib.effect_defaults(effect_type)
    e.props.reverb = p
    return e


def effects_equal(a, b):
    """Effect::are_equal on normalised effects: compare the live member only."""
    if a.type != b.type:
        return False
    if a.type == desc.NULL:
        return True
    m = desc.PROPS_MEMBER[a.type]
    return bytes(getattr(a.props, m)) == bytes(getattr(b.props, m))


class OracleApi:
    def __init__(self, channel_format, rate, effect_count):
        self.format, self.rate, self.effect_count = channel_format, rate, effect_count
        self.channels = desc.FORMAT_CHANNELS[channel_format]
        self.oracle = orc.Oracle(self.channels, effect_count)
        null = lib.effect_defaults(desc.NULL)
        self.deferred = [desc.Effect.from_buffer_copy(bytes(null)) for _ in range(effect_count)]
        self.active = [desc.Effect.from_buffer_copy(bytes(null)) for _ in range(effect_count)]
        self.params = [None] * effect_count
        self.seq = [0] * effect_count
        self.slot_changed = [True] * effect_count
        self.slot_restart = [True] * effect_count
        self.direct = desc.SendProps(1.0, 1.0, 1.0)
        self.direct_deferred = desc.SendProps(1.0, 1.0, 1.0)
        self.aux = [desc.SendProps(1.0, 1.0, 1.0) for _ in range(effect_count)]
        self.source_changed = True

    # ---- deferred setters ----
    def set_effect(self, slot, effect):
        self.deferred[slot] = desc.Effect.from_buffer_copy(bytes(effect))

    def set_effect_type(self, slot, effect_type):
        self.deferred[slot] = lib.effect_defaults(effect_type)

    def set_send_props(self, slot, gain, gain_hf, gain_lf):
        if slot < 0:
            self.direct_deferred = desc.SendProps(gain, gain_hf, gain_lf)
        else:
            # the reference writes the active aux props directly, un-normalised (src/oalsfxpp.cpp:3728-3733)
            self.aux[slot] = desc.SendProps(gain, gain_hf, gain_lf)

    def apply_changes(self):
        for i in range(self.effect_count):
            self.deferred[i] = lib.effect_normalized(self.deferred[i])
            if not effects_equal(self.deferred[i], self.active[i]):
                if self.deferred[i].type != self.active[i].type:
                    self.slot_restart[i] = True
                self.active[i] = desc.Effect.from_buffer_copy(bytes(self.deferred[i]))
                self.slot_changed[i] = True
        d = self.direct_deferred
        d = desc.SendProps(min(1.0, max(0.0, d.gain)), min(1.0, max(0.0, d.gain_hf)), min(1.0, max(0.0, d.gain_lf)))
        self.direct_deferred = d
        if bytes(d) != bytes(self.direct):
            self.direct = d
            self.source_changed = True
        for a in self.aux:
            if bytes(a) != bytes(desc.SendProps(1.0, 1.0, 1.0)):
                self.source_changed = True

    # ---- what mix_data's lazy refresh does ----
    def refresh(self):
        updated = False
        for i in range(self.effect_count):
            if not self.slot_changed[i]:
                continue
            self.slot_changed[i] = False
            updated = True
            p = lib.derive_slot(self.format, self.rate, self.active[i])
            self.seq[i] += 1
            p.update_seq = self.seq[i]
            self.params[i] = p
            self.oracle.set_slot(i, p, self.slot_restart[i])
            self.slot_restart[i] = False
        if self.source_changed:
            self.source_changed = False
            updated = True
        if updated:
            self.source_params = lib.derive_source(self.format, self.rate, self.direct, self.aux, [e.type for e in self.active])
            self.oracle.set_source(self.source_params)

    def mix(self, src):
        self.refresh()
        return self.oracle.mix(src)


def same_bits(a, b):
    """Bit-exact comparison of two float32 arrays; NaNs compare equal whatever their sign/payload
    (x86 and gfx950 encode the default NaN differently)."""
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
    b = np.ascontiguousarray(b, dtype=np.float32).reshape(-1)
    if a.size != b.size:
        return False, a.size
    bad = (a.view(np.uint32) != b.view(np.uint32)) & ~(np.isnan(a) & np.isnan(b))
    return not bad.any(), int(bad.sum())


def noise(seed, frames, channels):
    """Deterministic uniform [-1, 1) test signal (the benchmark generator)."""
    return orc.synth(seed, 0, frames * channels).reshape(frames, channels)


def struct_diff(a, b, path=""):
    """Field-by-field differences of two ctypes structures (for readable assertion messages)."""
    out = []
    if isinstance(a, (C.Structure, C.Union)):
        for name, *_ in a._fields_:
            out += struct_diff(getattr(a, name), getattr(b, name), f"{path}.{name}")
    elif isinstance(a, C.Array):
        for i in range(len(a)):
            out += struct_diff(a[i], b[i], f"{path}[{i}]")
    else:
        same = (a == b) or (isinstance(a, float) and (np.float32(a).tobytes() == np.float32(b).tobytes() or (a != a and b != b)))
        if not same:
            out.append(f"{path}: {a!r} != {b!r}")
    return out


class OracleShadow:
    """CPU oracle that follows one instance of a `Batch`: it is fed the descriptors the batch derived
    (read back through the C ABI), so outputs, states and rings can be compared word for word."""

    def __init__(self, batch, instance):
        self.batch, self.instance = batch, instance
        self.oracle = orc.Oracle(batch.channels, batch.effect_count)
        self.seq = [None] * batch.effect_count
        self.types = [None] * batch.effect_count
        self.source_raw = None

    def sync(self):
        for s in range(self.batch.effect_count):
            p, _ = self.batch.read_slot(self.instance, s)
            if self.seq[s] != p.update_seq:
                self.oracle.set_slot(s, p, restart=(self.types[s] != p.type))
                self.seq[s], self.types[s] = p.update_seq, p.type
        sp, _ = self.batch.read_source(self.instance)
        if bytes(sp) != self.source_raw:
            self.oracle.set_source(sp)
            self.source_raw = bytes(sp)

    def mix(self, src):
        self.sync()
        return self.oracle.mix(src)

    def compare_state(self):
        """Differences between the batch's device state / rings and the oracle's, as strings."""
        self.sync()  # applied-but-not-yet-mixed changes (e.g. a type change re-creates the state) reach the oracle first
        diffs = []
        for s in range(self.batch.effect_count):
            p, st = self.batch.read_slot(self.instance, s)
            if p.type in desc.STATE_MEMBER:
                m = desc.STATE_MEMBER[p.type]
                diffs += struct_diff(getattr(st.u, m), getattr(self.oracle.state(s).u, m), f"slot{s}.{m}")
            ring_gpu = self.batch.read_ring(self.instance, s)
            ring_cpu = self.oracle.ring(s)
            ok, nbad = same_bits(ring_gpu, ring_cpu)
            if not ok:
                diffs.append(f"slot{s}.ring: {nbad} words differ (sizes {ring_gpu.size}/{ring_cpu.size})")
        _, sst = self.batch.read_source(self.instance)
        diffs += struct_diff(sst, self.oracle.source_state(), "source_state")
        return diffs


class ShadowArmy:
    """One CPU oracle per listed instance of a `Batch` (all of them by default), mixed on a thread pool: the oracle's entry
    points are plain C calls, which ctypes runs without the interpreter lock.  For checks that follow *every* instance of a
    full-size batch instead of a sample."""

    def __init__(self, batch, instances=None, threads=None):
        from concurrent.futures import ThreadPoolExecutor
        self.batch = batch
        self.instances = list(range(batch.n)) if instances is None else list(instances)
        self.shadows = [OracleShadow(batch, i) for i in self.instances]
        self.pool = ThreadPoolExecutor(max_workers=threads or min(16, len(os.sched_getaffinity(0))))

    def sync(self):
        for s in self.shadows:
            s.sync()

    def mix(self, x):
        """x: the batch's whole input [n][frames][channels]; returns the oracle outputs of the followed instances, in list order."""
        self.sync()   # (descriptor read-backs are HIP calls: kept on the calling thread)
        return np.stack(list(self.pool.map(lambda s: s.oracle.mix(x[s.instance]), self.shadows)))

    def differing(self, y, ref):
        """Followed instances whose device output `y[instance]` differs from the oracle's `ref[k]`: [(instance, samples differing)]."""
        got = np.ascontiguousarray(y[self.instances]).view(np.uint32)
        want = np.ascontiguousarray(ref).view(np.uint32)
        nan = np.isnan(y[self.instances]) & np.isnan(ref)
        bad = ((got != want) & ~nan).reshape(len(self.instances), -1).sum(axis=1)
        return [(self.instances[k], int(bad[k])) for k in np.nonzero(bad)[0]]


def crossfade_followable(before, after):
    """Mirror of the host's hint (hip/batch.cpp: crossfade_followable) on two `desc.ReverbParams`: can the cross-fading build of the
    steady-state reverb kernel take an instance from one to the other (both tap sets ones its most general build accepts)?"""
    sway = 0
    if before.mod_depth != 0.0 or after.mod_depth != 0.0:
        sway = 1 + int(max(abs(before.mod_depth), abs(after.mod_depth)))
    for p in (before, after):
        for j in range(4):
            if (p.early_tap[j] < 0 or p.early_ap_off[j] < 4 or p.early_line_off[j] < 0 or p.late_tap[j] < after.late_feed_tap
                    or p.late_ap_off[j] < 4 or p.late_line_off[j] < 64 + sway):
                return False
    return True


def reverb_params(effect, fmt=desc.FMT_STEREO, rate=48000):
    return lib.derive_slot(fmt, rate, lib.effect_normalized(effect)).u.reverb


def steady_build(symbol):
    """What `Batch.last_reverb_kernel` says about the steady-state launch: {'kinds': the grid of several kinds, 'fp': a proven-steady build
    (no steady-state test inside), 'xf': the build that follows property changes}.  Template arguments of k_reverb_steady_coop: channels,
    wavefronts, TL, HY, MD, ST, RG, FP, XF, NF, SF, CR ('cr': 2 every ring line written in whole cache lines, the build for write positions off the line grid)."""
    if symbol.startswith("k_reverb_steady_kinds"):
        args = [a.strip() for a in symbol[symbol.index("<") + 1: symbol.rindex(">")].split(",")]
        return {"kinds": True, "fp": False, "xf": False, "rg": False, "cr": int(args[3]) if len(args) > 3 else 0}
    args = [a.strip() for a in symbol[symbol.index("<") + 1: symbol.rindex(">")].split(",")]
    flag = lambda k: len(args) > k and args[k] == "true"
    return {"kinds": False, "fp": flag(7), "xf": flag(8), "rg": flag(6), "cr": int(args[11]) if len(args) > 11 else 0}
