"""Python host mirror of the batch C ABI (include/oalsfx_hip.h).

`Batch` advances N independent effect chains with one call; `Api` is the one-instance view with the
method names of the reference's `oalsfxpp::Api` (reference src/oalsfxpp.h:760-922).  Both are thin:
every call goes straight to liboalsfx_hip.so, nothing is computed in Python.
"""
import ctypes as C
import weakref

import numpy as np

from . import desc, lib

_fp = C.POINTER(C.c_float)


class BatchError(RuntimeError):
    pass


class Batch:
    def __init__(self, n_instances, channel_format=desc.FMT_STEREO, sampling_rate=48000, effect_count=1, device_id=0):
        self._lib = lib.load()
        h = self._lib.oalsfx_batch_create(n_instances, channel_format, sampling_rate, effect_count, device_id)
        if not h:
            raise BatchError(self._lib.oalsfx_last_error().decode())
        self._h = C.c_void_p(h)
        self.n = n_instances
        self.channel_format = channel_format
        self.rate = sampling_rate
        self.effect_count = effect_count
        self.channels = self._lib.oalsfx_batch_channels(self._h)
        self.device_id = device_id

    def close(self):
        if getattr(self, "_h", None):
            self._lib.oalsfx_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, ok):
        if not ok:
            raise BatchError(self._lib.oalsfx_batch_error(self._h).decode())

    @property
    def error(self):
        return self._lib.oalsfx_batch_error(self._h).decode()

    # ---- deferred setters (instance range [first, first+count)) ----
    def _range(self, first, count):
        return first, (self.n - first if count is None else count)

    def set_effect(self, slot, effect, first=0, count=None):
        """One `desc.Effect` broadcast to the range, or a sequence of them (one per instance)."""
        first, count = self._range(first, count)
        if isinstance(effect, desc.Effect):
            self._check(self._lib.oalsfx_batch_set_effect(self._h, first, count, slot, C.byref(effect), 0))
        else:
            arr = (desc.Effect * count)(*effect)
            self._check(self._lib.oalsfx_batch_set_effect(self._h, first, count, slot, arr, C.sizeof(desc.Effect)))

    def set_effect_at(self, slot, instances, effects):
        """`effects[k]` (or the one `desc.Effect`) for instance `instances[k]`: one foreign call for instances that are not neighbours."""
        idx = (C.c_int * len(instances))(*instances)
        if isinstance(effects, desc.Effect):
            self._check(self._lib.oalsfx_batch_set_effect_at(self._h, idx, len(instances), slot, C.byref(effects), 0))
        else:
            arr = (desc.Effect * len(instances))(*effects)
            self._check(self._lib.oalsfx_batch_set_effect_at(self._h, idx, len(instances), slot, arr, C.sizeof(desc.Effect)))

    def set_effect_type(self, slot, effect_type, first=0, count=None):
        first, count = self._range(first, count)
        self._check(self._lib.oalsfx_batch_set_effect_type(self._h, first, count, slot, effect_type))

    def set_effect_props(self, slot, props_union, first=0, count=None):
        first, count = self._range(first, count)
        self._check(self._lib.oalsfx_batch_set_effect_props(self._h, first, count, slot, C.byref(props_union), 0))

    def set_send_props(self, slot, gain, gain_hf, gain_lf, first=0, count=None):
        first, count = self._range(first, count)
        p = desc.SendProps(gain, gain_hf, gain_lf)
        self._check(self._lib.oalsfx_batch_set_send_props(self._h, first, count, slot, C.byref(p)))

    def get_effect(self, instance, slot, deferred=False):
        e = desc.Effect()
        self._check(self._lib.oalsfx_batch_get_effect(self._h, instance, slot, 1 if deferred else 0, C.byref(e)))
        return e

    def get_send_props(self, instance, slot, deferred=False):
        p = desc.SendProps()
        self._check(self._lib.oalsfx_batch_get_send_props(self._h, instance, slot, 1 if deferred else 0, C.byref(p)))
        return p

    def apply_changes(self, first=0, count=None):
        first, count = self._range(first, count)
        self._check(self._lib.oalsfx_batch_apply_changes(self._h, first, count))

    # ---- the hot path ----
    def mix(self, src):
        """src: float32 array [n][frames][channels] on the host; returns the same shape."""
        src = np.ascontiguousarray(src, dtype=np.float32)
        assert src.ndim == 3 and src.shape[0] == self.n and src.shape[2] == self.channels, src.shape
        dst = np.empty_like(src)
        self._check(self._lib.oalsfx_batch_mix(self._h, src.shape[1], src.ctypes.data_as(_fp), dst.ctypes.data_as(_fp)))
        return dst

    def mix_async(self, src, dst):
        """Queues one buffer from / to host arrays [n][frames][channels] (page-locked ones overlap with the kernels) and returns;
        both must stay untouched until wait() or until three more mix_async calls have returned."""
        assert src.dtype == np.float32 and dst.dtype == np.float32 and src.flags.c_contiguous and dst.flags.c_contiguous and src.shape == dst.shape
        assert src.ndim == 3 and src.shape[0] == self.n and src.shape[2] == self.channels, src.shape
        self._check(self._lib.oalsfx_batch_mix_async(self._h, src.shape[1], src.ctypes.data_as(_fp), dst.ctypes.data_as(_fp)))

    def wait(self):
        self._check(self._lib.oalsfx_batch_wait(self._h))

    def pinned_array(self, frames):
        """A page-locked float32 array [n][frames][channels].  The allocation belongs to the array, not to the Batch: it is released
        (oalsfx_pinned_free) when the array and every view of it are gone, so an array may outlive close(), and a loop that asks for a
        fresh array per step holds only what it still references.  (Wait for the calls in flight -- wait() -- before dropping one.)"""
        count = self.n * frames * self.channels
        p = self._lib.oalsfx_pinned_alloc(count * 4)
        if not p:
            raise BatchError("oalsfx_pinned_alloc failed")
        buf = (C.c_float * count).from_address(p)
        # numpy keeps `buf` alive as the base of the array and of its views; the finalizer holds the library handle
        weakref.finalize(buf, self._lib.oalsfx_pinned_free, C.c_void_p(p))
        return np.frombuffer(buf, dtype=np.float32).reshape(self.n, frames, self.channels)

    def mix_device(self, frames, src_ptr, dst_ptr, stream=None):
        """Buffers already in device memory (raw addresses, e.g. torch.Tensor.data_ptr()); asynchronous."""
        self._check(self._lib.oalsfx_batch_mix_device(self._h, frames, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), C.c_void_p(stream or 0)))

    def synchronize(self):
        self._check(self._lib.oalsfx_batch_synchronize(self._h))

    @property
    def stream(self):
        return self._lib.oalsfx_batch_stream(self._h)

    def fill_synthetic(self, frames, buffer_index, dst_ptr, stream=None):
        self._check(self._lib.oalsfx_batch_fill_synthetic(self._h, frames, buffer_index, C.c_void_p(dst_ptr), C.c_void_p(stream or 0)))

    # ---- read-back ----
    def read_slot(self, instance, slot):
        p, s = desc.SlotParams(), desc.SlotState()
        self._check(self._lib.oalsfx_batch_read_slot(self._h, instance, slot, C.byref(p), C.byref(s)))
        return p, s

    def read_ring(self, instance, slot):
        n = self._lib.oalsfx_batch_read_ring(self._h, instance, slot, None, 0)
        out = np.zeros(n, dtype=np.float32)
        if n:
            self._lib.oalsfx_batch_read_ring(self._h, instance, slot, out.ctypes.data_as(_fp), n)
        return out

    def read_source(self, instance):
        p, s = desc.SourceParams(), desc.SourceState()
        self._check(self._lib.oalsfx_batch_read_source(self._h, instance, C.byref(p), C.byref(s)))
        return p, s

    # ---- kernel timing (HIP events on the launch stream) ----
    def kernel_timing(self, enable=1):
        """0 / False: off; 1 / True: every mix call carries timing events; k > 1: every k-th call."""
        self._check(self._lib.oalsfx_batch_kernel_timing(self._h, int(enable)))

    def event_overhead(self, repeats=100):
        """Average microseconds an event pair around an empty kernel reads on the batch's stream."""
        us = C.c_double(0.0)
        self._check(self._lib.oalsfx_batch_event_overhead(self._h, repeats, C.byref(us)))
        return us.value

    def kernel_timing_samples(self, effect_type, max_samples=4096):
        """Per-launch durations (microseconds, raw event pairs) of the timed launches of `effect_type`."""
        buf = (C.c_double * max_samples)()
        n = self._lib.oalsfx_batch_kernel_timing_samples(self._h, effect_type, buf, max_samples)
        self._check(n >= 0)
        return list(buf[:min(n, max_samples)])

    def plan(self, slot=0):
        """(ring-light, reverbs proven steady, reverbs believed steady, reverbs on the general kernel) for the next mix call."""
        c = (C.c_int * 4)()
        self._check(self._lib.oalsfx_batch_plan(self._h, slot, c))
        return tuple(c)

    def placement(self):
        """(chunks, candidates probed, probe us on the chunk kept, probe us on the slowest candidate seen) of the delay-line placement search."""
        c, k = C.c_int(0), C.c_int(0)
        a, w = C.c_double(0.0), C.c_double(0.0)
        self._lib.oalsfx_batch_placement(self._h, C.byref(c), C.byref(k), C.byref(a), C.byref(w))
        return c.value, k.value, a.value, w.value

    @property
    def chained_calls(self):
        """mix_device calls (own stream) that overlapped with their neighbours on the device so far."""
        return self._lib.oalsfx_batch_chained_calls(self._h)

    def host_pipeline(self):
        """(form, probe us per call on three streams, probe us per call on one stream) of mix_async; form 0: still probing."""
        f, a, c = C.c_int(0), C.c_double(0), C.c_double(0)
        self._lib.oalsfx_debug_host_pipeline(self._h, C.byref(f), C.byref(a), C.byref(c))
        return f.value, a.value, c.value

    def chain_started(self):
        """(host count, device count) of the workgroups of chained launches started so far: the two must agree.  Waits."""
        h, d = C.c_uint(0), C.c_uint(0)
        self._check(self._lib.oalsfx_debug_chain_started(self._h, C.byref(h), C.byref(d)))
        return h.value, d.value

    @property
    def last_reverb_kernel(self):
        return (self._lib.oalsfx_batch_last_reverb_kernel(self._h) or b"").decode()

    def kernel_timing_read(self, effect_type):
        n, ms = C.c_int(0), C.c_double(0.0)
        self._check(self._lib.oalsfx_batch_kernel_timing_read(self._h, effect_type, C.byref(n), C.byref(ms)))
        return n.value, ms.value


class Api:
    """One effect chain with the reference's method names; a `Batch` of one instance underneath."""

    def __init__(self):
        self._b = None
        self._error = ""

    def initialize(self, channel_format, sampling_rate, effect_count, device_id=0):
        self.uninitialize()
        try:
            self._b = Batch(1, channel_format, sampling_rate, effect_count, device_id)
        except BatchError as e:
            self._error = str(e)
            return False
        return True

    def is_initialized(self):
        return self._b is not None

    def uninitialize(self):
        if self._b is not None:
            self._b.close()
            self._b = None

    def get_error_message(self):
        return self._error

    def _guard(self, fn, fail=False):
        if self._b is None:
            self._error = "Not initialized."
            return fail
        try:
            return fn()
        except BatchError as e:
            self._error = str(e)
            return fail

    def get_sampling_rate(self):
        return self._guard(lambda: self._b.rate, 0)

    def get_channel_format(self):
        return self._guard(lambda: self._b.channel_format, desc.FMT_NONE)

    def get_channel_count(self):
        return self._guard(lambda: self._b.channels, 0)

    def get_effect_count(self):
        return self._guard(lambda: self._b.effect_count, 0)

    def set_effect_type(self, index, effect_type):
        return self._guard(lambda: self._b.set_effect_type(index, effect_type) or True)

    def set_effect(self, index, effect):
        # the reference's Api::set_effect stores the effect and returns false (src/oalsfxpp.cpp:3655-3657)
        self._guard(lambda: self._b.set_effect(index, effect))
        return False

    def get_effect(self, index, deferred=False):
        return self._guard(lambda: self._b.get_effect(0, index, deferred), None)

    def set_send_props(self, index, gain, gain_hf, gain_lf):
        return self._guard(lambda: self._b.set_send_props(index, gain, gain_hf, gain_lf) or True)

    def apply_changes(self):
        return self._guard(lambda: self._b.apply_changes() or True)

    def mix(self, src):
        """src: [frames][channels] float32; returns the mixed frames or None on failure."""
        src = np.ascontiguousarray(src, dtype=np.float32)
        if src.size == 0:
            return src.copy()
        return self._guard(lambda: self._b.mix(src[None])[0], None)


def trim_pools():
    """Gives the uncached device memory that waits for reuse back to the runtime (oalsfx_trim_pools); returns the bytes freed."""
    return lib.load().oalsfx_trim_pools()


def pools_waiting_bytes():
    return lib.load().oalsfx_pools_waiting_bytes()


class Group:
    """One instance range over several devices (oalsfx_group_*): a batch and a host thread per device, contiguous shards."""

    def __init__(self, n_total, device_ids, channel_format=desc.FMT_STEREO, sampling_rate=48000, effect_count=1):
        self._lib = lib.load()
        ids = (C.c_int * len(device_ids))(*device_ids)
        h = self._lib.oalsfx_group_create(n_total, ids, len(device_ids), channel_format, sampling_rate, effect_count)
        if not h:
            raise BatchError(self._lib.oalsfx_group_last_error().decode())
        self._h = C.c_void_p(h)
        self.n = n_total
        self.channels = self._lib.oalsfx_group_channels(self._h)
        self.shards = []
        for k in range(self._lib.oalsfx_group_devices(self._h)):
            d, f, c = C.c_int(), C.c_int(), C.c_int()
            self._lib.oalsfx_group_shard(self._h, k, C.byref(d), C.byref(f), C.byref(c))
            self.shards.append((d.value, f.value, c.value))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.oalsfx_group_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, ok):
        if not ok:
            raise BatchError(self._lib.oalsfx_group_error(self._h).decode())

    def set_effect_type(self, slot, effect_type, first=0, count=None):
        self._check(self._lib.oalsfx_group_set_effect_type(self._h, first, self.n - first if count is None else count, slot, effect_type))

    def set_effect(self, slot, effects, first=0):
        """effects: one desc.Effect for the whole range from `first` on, or a list of them (one per instance from `first` on)."""
        if isinstance(effects, desc.Effect):
            self._check(self._lib.oalsfx_group_set_effect(self._h, first, self.n - first, slot, C.byref(effects), 0))
        else:
            arr = (desc.Effect * len(effects))(*effects)
            self._check(self._lib.oalsfx_group_set_effect(self._h, first, len(effects), slot, arr, C.sizeof(desc.Effect)))

    def set_send_props(self, slot, gain, gain_hf, gain_lf, first=0, count=None):
        sp = desc.SendProps(gain, gain_hf, gain_lf)
        self._check(self._lib.oalsfx_group_set_send_props(self._h, first, self.n - first if count is None else count, slot, C.byref(sp)))

    def apply_changes(self, first=0, count=None):
        self._check(self._lib.oalsfx_group_apply_changes(self._h, first, self.n - first if count is None else count))

    def mix(self, src):
        """src: float32 [n_total][frames][channels] in host memory; returns the output array."""
        src = np.ascontiguousarray(src, dtype=np.float32)
        frames = src.shape[1]
        dst = np.empty_like(src)
        self._check(self._lib.oalsfx_group_mix(self._h, frames, src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p)))
        return dst

    def mix_device(self, frames, src_ptrs, dst_ptrs):
        """One device pointer per shard each (that shard's [count][frames][channels]); queues and returns."""
        s = (C.c_void_p * len(src_ptrs))(*src_ptrs)
        d = (C.c_void_p * len(dst_ptrs))(*dst_ptrs)
        self._check(self._lib.oalsfx_group_mix_device(self._h, frames, s, d))

    def synchronize(self):
        self._check(self._lib.oalsfx_group_synchronize(self._h))
