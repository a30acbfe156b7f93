"""TEST INFRASTRUCTURE: ctypes access to the parity checkers.

  * `Oracle`    -- oracle/liboracle.so, the CPU restatement of the process path (travels to the GPU box)
  * `Reference` -- oracle/_ref/libref.so, the untouched reference compiled in the build container
                   (absent on the GPU box; `have_reference()` says whether it is there)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from oalsfxpp_amd import desc

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "liboracle.so")
_ORACLE_FAST_SO = os.path.join(_HERE, "liboracle_fast.so")  # -O3 -march=x86-64-v3, FMA contraction allowed: cpu_baseline only
_REF_SO = os.path.join(_HERE, "_ref", "libref.so")

_fp = C.POINTER(C.c_float)


def build(ref=True):
    """(Re)build the checkers with oracle/Makefile; the reference part is skipped when /root/reference is absent."""
    targets = ["oracle", "fast"] + (["ref"] if ref else [])
    subprocess.run(["make", "-C", _HERE] + targets, check=True, stdout=subprocess.DEVNULL)


def have_reference():
    return os.path.exists(_REF_SO)


def _load(path):
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not built: run `make -C oracle`")
    return C.CDLL(path)


_oracle = {}


def oracle_lib(fast=False):
    """fast=False: the parity build (bit-identical to the reference); fast=True: the tolerance-checked speed build."""
    global _oracle
    if fast not in _oracle:
        lib = _load(_ORACLE_FAST_SO if fast else _ORACLE_SO)
        lib.oracle_create.restype = C.c_void_p
        lib.oracle_create.argtypes = [C.c_int, C.c_int]
        lib.oracle_destroy.argtypes = [C.c_void_p]
        lib.oracle_set_source.argtypes = [C.c_void_p, C.POINTER(desc.SourceParams)]
        lib.oracle_set_slot.argtypes = [C.c_void_p, C.c_int, C.POINTER(desc.SlotParams), C.c_int]
        lib.oracle_mix.argtypes = [C.c_void_p, C.c_int, _fp, _fp]
        lib.oracle_get_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(desc.SlotState)]
        lib.oracle_get_source_state.argtypes = [C.c_void_p, C.POINTER(desc.SourceState)]
        lib.oracle_get_ring.restype = C.c_int
        lib.oracle_get_ring.argtypes = [C.c_void_p, C.c_int, _fp, C.c_int]
        lib.oracle_synth.argtypes = [C.c_uint32, C.c_uint32, C.c_int, _fp]
        lib.oracle_bench.restype = C.c_double
        lib.oracle_bench.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        _oracle[fast] = lib
    return _oracle[fast]


def synth(instance, buffer_index, count):
    """The benchmark's synthetic input: xorshift32 per (instance, buffer), uniform [-1, 1)."""
    out = np.empty(count, dtype=np.float32)
    oracle_lib().oracle_synth(instance, buffer_index, count, out.ctypes.data_as(_fp))
    return out


class Oracle:
    """One instance of the CPU restatement, driven with descriptors from the host update path."""

    def __init__(self, channels, slots, fast=False):
        self.lib = oracle_lib(fast)
        self.channels, self.slots = channels, slots
        self.h = C.c_void_p(self.lib.oracle_create(channels, slots))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oracle_destroy(self.h)
            self.h = None

    def set_source(self, params):
        self.lib.oracle_set_source(self.h, C.byref(params))

    def set_slot(self, slot, params, restart):
        self.lib.oracle_set_slot(self.h, slot, C.byref(params), 1 if restart else 0)

    def mix(self, src):
        src = np.ascontiguousarray(src, dtype=np.float32).reshape(-1)
        frames = src.size // self.channels
        dst = np.empty_like(src)
        self.lib.oracle_mix(self.h, frames, src.ctypes.data_as(_fp), dst.ctypes.data_as(_fp))
        return dst.reshape(frames, self.channels)

    def state(self, slot):
        s = desc.SlotState()
        self.lib.oracle_get_state(self.h, slot, C.byref(s))
        return s

    def source_state(self):
        s = desc.SourceState()
        self.lib.oracle_get_source_state(self.h, C.byref(s))
        return s

    def ring(self, slot):
        n = self.lib.oracle_get_ring(self.h, slot, None, 0)
        out = np.zeros(n, dtype=np.float32)
        if n:
            self.lib.oracle_get_ring(self.h, slot, out.ctypes.data_as(_fp), n)
        return out

    def bench(self, n_instances, frames, warmup, buffers, threads):
        return self.lib.oracle_bench(self.h, n_instances, frames, warmup, buffers, threads)


_ref = None


def ref_lib():
    global _ref
    if _ref is None:
        lib = _load(_REF_SO)
        lib.ref_create.restype = C.c_void_p
        lib.ref_create.argtypes = [C.c_int, C.c_int, C.c_int]
        lib.ref_destroy.argtypes = [C.c_void_p]
        for name in ("ref_set_effect", "ref_set_effect_props"):
            getattr(lib, name).argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        lib.ref_set_effect_type.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.ref_get_effect.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        lib.ref_set_send_props.argtypes = [C.c_void_p, C.c_int, _fp]
        lib.ref_get_send_props.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp]
        lib.ref_apply_changes.argtypes = [C.c_void_p]
        lib.ref_mix.argtypes = [C.c_void_p, C.c_int, _fp, _fp]
        lib.ref_error.restype = C.c_char_p
        lib.ref_error.argtypes = [C.c_void_p]
        if hasattr(lib, "ref_bench"):
            lib.ref_bench.restype = C.c_double
            lib.ref_bench.argtypes = [C.c_int, C.c_int, C.c_void_p] + [C.c_int] * 5
        lib.ref_effect_defaults.argtypes = [C.c_int, C.c_void_p]
        lib.ref_effect_normalize.argtypes = [C.c_void_p]
        lib.ref_preset_name.restype = C.c_char_p
        lib.ref_preset_props.argtypes = [C.c_int, C.c_void_p]
        lib.ref_refresh.argtypes = [C.c_void_p]
        lib.ref_channel_count.argtypes = [C.c_void_p]
        lib.ref_dump_source.argtypes = [C.c_void_p, C.POINTER(desc.SourceParams), C.POINTER(desc.SourceState)]
        lib.ref_dump_slot.argtypes = [C.c_void_p, C.c_int, C.POINTER(desc.SlotParams), C.POINTER(desc.SlotState)]
        lib.ref_dump_ring.argtypes = [C.c_void_p, C.c_int, C.c_int, _fp]
        _ref = lib
    return _ref


class Reference:
    """One oalsfxpp::Api of the compiled reference."""

    def __init__(self, channel_format, rate, effect_count):
        self.lib = ref_lib()
        h = self.lib.ref_create(channel_format, rate, effect_count)
        if not h:
            raise ValueError("reference Api::initialize failed")
        self.h = C.c_void_p(h)
        self.channels = self.lib.ref_channel_count(self.h)
        self.effect_count = effect_count

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_destroy(self.h)
            self.h = None

    def set_effect(self, slot, effect):
        return self.lib.ref_set_effect(self.h, slot, C.byref(effect))

    def set_effect_type(self, slot, effect_type):
        return self.lib.ref_set_effect_type(self.h, slot, effect_type)

    def set_effect_props(self, slot, props_union):
        return self.lib.ref_set_effect_props(self.h, slot, C.byref(props_union))

    def get_effect(self, slot, deferred=False):
        e = desc.Effect()
        ok = self.lib.ref_get_effect(self.h, slot, 1 if deferred else 0, C.byref(e))
        return ok, e

    def set_send_props(self, slot, gain, gain_hf, gain_lf):
        p = (C.c_float * 3)(gain, gain_hf, gain_lf)
        return self.lib.ref_set_send_props(self.h, slot, p)

    def get_send_props(self, slot, deferred=False):
        p = (C.c_float * 3)()
        ok = self.lib.ref_get_send_props(self.h, slot, 1 if deferred else 0, p)
        return ok, tuple(p)

    def apply_changes(self):
        return self.lib.ref_apply_changes(self.h)

    def mix(self, src):
        src = np.ascontiguousarray(src, dtype=np.float32).reshape(-1)
        frames = src.size // self.channels
        dst = np.empty_like(src)
        ok = self.lib.ref_mix(self.h, frames, src.ctypes.data_as(_fp), dst.ctypes.data_as(_fp))
        assert ok, self.lib.ref_error(self.h)
        return dst.reshape(frames, self.channels)

    def refresh(self):
        self.lib.ref_refresh(self.h)

    def dump_source(self):
        p, s = desc.SourceParams(), desc.SourceState()
        self.lib.ref_dump_source(self.h, C.byref(p), C.byref(s))
        return p, s

    def dump_slot(self, slot):
        p, s = desc.SlotParams(), desc.SlotState()
        self.lib.ref_dump_slot(self.h, slot, C.byref(p), C.byref(s))
        return p, s

    def dump_rings(self, slot, params):
        """All rings of the slot concatenated in the repository's slab layout."""
        t = params.type
        if t in (desc.CHORUS, desc.FLANGER):
            out = np.zeros(2 * params.u.moddelay.ring_len, dtype=np.float32)
            self.lib.ref_dump_ring(self.h, slot, 0, out.ctypes.data_as(_fp))
            return out
        if t == desc.ECHO:
            out = np.zeros(params.u.echo.ring_len, dtype=np.float32)
            self.lib.ref_dump_ring(self.h, slot, 0, out.ctypes.data_as(_fp))
            return out
        if t in (desc.REVERB, desc.EAX_REVERB):
            rp = params.u.reverb
            out = np.zeros(4 * sum(rp.ring_len), dtype=np.float32)
            for r in range(5):
                seg = out[rp.ring_off[r]: rp.ring_off[r] + 4 * rp.ring_len[r]]
                self.lib.ref_dump_ring(self.h, slot, r, seg.ctypes.data_as(_fp))
            return out
        return np.zeros(0, dtype=np.float32)


def ref_effect_defaults(effect_type):
    e = desc.Effect()
    ref_lib().ref_effect_defaults(effect_type, C.byref(e))
    return e


def ref_preset(index):
    p = desc.ReverbProps()
    ref_lib().ref_preset_props(index, C.byref(p))
    return ref_lib().ref_preset_name(index).decode(), p
