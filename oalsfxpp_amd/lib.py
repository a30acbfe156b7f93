"""Loader for liboalsfx_hip.so (the C ABI of include/oalsfx_hip.h).

The library is built in-tree by `oalsfxpp_amd.build.build_all()` (hipcc, gfx950).  There is no
Python or CPU fallback: if the shared object is missing, loading fails loudly.
"""
import ctypes as C
import os

from . import desc

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liboalsfx_hip.so")

_fp = C.POINTER(C.c_float)
_lib = None

# name -> (restype, argtypes); every symbol include/oalsfx_hip.h and include/oalsfx_hip_debug.h declare
SIGNATURES = {
    "oalsfx_batch_create": (C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "oalsfx_batch_destroy": (None, [C.c_void_p]),
    "oalsfx_batch_error": (C.c_char_p, [C.c_void_p]),
    "oalsfx_last_error": (C.c_char_p, []),
    "oalsfx_batch_instances": (C.c_int, [C.c_void_p]),
    "oalsfx_batch_channels": (C.c_int, [C.c_void_p]),
    "oalsfx_batch_sampling_rate": (C.c_int, [C.c_void_p]),
    "oalsfx_batch_effect_count": (C.c_int, [C.c_void_p]),
    "oalsfx_batch_set_effect": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "oalsfx_batch_set_effect_at": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "oalsfx_batch_set_effect_type": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "oalsfx_batch_set_effect_props": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "oalsfx_batch_set_send_props": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(desc.SendProps)]),
    "oalsfx_batch_get_effect": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(desc.Effect)]),
    "oalsfx_batch_get_send_props": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(desc.SendProps)]),
    "oalsfx_batch_apply_changes": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "oalsfx_batch_mix": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp]),
    "oalsfx_batch_mix_timed": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp, C.POINTER(C.c_double)]),
    "oalsfx_batch_mix_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "oalsfx_batch_synchronize": (C.c_int, [C.c_void_p]),
    "oalsfx_batch_mix_async": (C.c_int, [C.c_void_p, C.c_int, _fp, _fp]),
    "oalsfx_batch_wait": (C.c_int, [C.c_void_p]),
    "oalsfx_pinned_alloc": (C.c_void_p, [C.c_ulonglong]),
    "oalsfx_pinned_free": (None, [C.c_void_p]),
    "oalsfx_batch_stream": (C.c_void_p, [C.c_void_p]),
    "oalsfx_batch_read_slot": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(desc.SlotParams), C.POINTER(desc.SlotState)]),
    "oalsfx_batch_read_ring": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _fp, C.c_int]),
    "oalsfx_batch_read_source": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(desc.SourceParams), C.POINTER(desc.SourceState)]),
    "oalsfx_batch_fill_synthetic": (C.c_int, [C.c_void_p, C.c_int, C.c_uint, C.c_void_p, C.c_void_p]),
    "oalsfx_batch_kernel_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "oalsfx_batch_kernel_timing_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "oalsfx_batch_event_overhead": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "oalsfx_batch_kernel_timing_samples": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int]),
    "oalsfx_batch_plan": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "oalsfx_batch_placement": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "oalsfx_batch_last_reverb_kernel": (C.c_char_p, [C.c_void_p]),
    "oalsfx_batch_mix_gather": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "oalsfx_batch_chained_calls": (C.c_longlong, [C.c_void_p]),
    "oalsfx_group_create": (C.c_void_p, [C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int]),
    "oalsfx_group_destroy": (None, [C.c_void_p]),
    "oalsfx_group_error": (C.c_char_p, [C.c_void_p]),
    "oalsfx_group_last_error": (C.c_char_p, []),
    "oalsfx_group_instances": (C.c_int, [C.c_void_p]),
    "oalsfx_group_channels": (C.c_int, [C.c_void_p]),
    "oalsfx_group_devices": (C.c_int, [C.c_void_p]),
    "oalsfx_group_shard": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "oalsfx_group_batch": (C.c_void_p, [C.c_void_p, C.c_int]),
    "oalsfx_group_set_effect": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "oalsfx_group_set_effect_type": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "oalsfx_group_set_effect_props": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "oalsfx_group_set_send_props": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(desc.SendProps)]),
    "oalsfx_group_apply_changes": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "oalsfx_group_mix": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "oalsfx_group_mix_device": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "oalsfx_group_synchronize": (C.c_int, [C.c_void_p]),
    "oalsfx_trim_pools": (C.c_ulonglong, []),
    "oalsfx_pools_waiting_bytes": (C.c_ulonglong, []),
    "oalsfx_debug_chain_same_cu": (C.c_longlong, [C.c_void_p]),
    "oalsfx_debug_chain_started": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]),
    "oalsfx_debug_gate_skew": (None, [C.c_void_p, C.c_uint]),
    "oalsfx_debug_chain_given_up": (C.c_int, [C.c_void_p]),
    "oalsfx_debug_host_pipeline": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "oalsfx_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "oalsfx_debug_set_flags": (None, [C.c_int]),
    "oalsfx_debug_ring_address": (C.c_ulonglong, [C.c_void_p, C.c_int, C.c_int]),
    "oalsfx_debug_move_rings": (C.c_int, [C.c_void_p, C.c_int]),
    "oalsfx_debug_probe_rings": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "oalsfx_debug_probe_pointer": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "oalsfx_debug_hbm_sweep": (C.c_int, [C.c_int, C.c_ulonglong, C.c_int, C.c_int]),
    "oalsfx_debug_stream_pattern": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "oalsfx_host_effect_defaults": (None, [C.c_int, C.POINTER(desc.Effect)]),
    "oalsfx_host_effect_normalize": (None, [C.POINTER(desc.Effect)]),
    "oalsfx_host_derive_slot": (C.c_int, [C.c_int, C.c_int, C.POINTER(desc.Effect), C.POINTER(desc.SlotParams)]),
    "oalsfx_host_derive_source": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(desc.SendProps), C.POINTER(desc.SendProps),
                                            C.POINTER(C.c_int), C.POINTER(desc.SourceParams)]),
    "oalsfx_host_ring_floats": (C.c_int, [C.c_int, C.c_int]),
    "oalsfx_host_channel_count": (C.c_int, [C.c_int]),
    "oalsfx_host_preset_count": (C.c_int, []),
    "oalsfx_host_preset_name": (C.c_char_p, [C.c_int]),
    "oalsfx_host_preset": (C.c_int, [C.c_int, C.c_void_p]),
}


def load(path=None):
    """Load the shared library and attach the prototypes.  Raises if the library is not built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("OALSFX_LIB") or LIB_PATH
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the process path.")
    lib = C.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if name.startswith("oalsfx_host_") or os.environ.get("OALSFX_LIB") is None:
                raise
            continue  # a host-only development build lacks the batch entry points
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


# ---- thin helpers over the host-only entry points ----
def effect_defaults(effect_type):
    e = desc.Effect()
    load().oalsfx_host_effect_defaults(effect_type, C.byref(e))
    return e


def effect_normalized(effect):
    e = desc.Effect.from_buffer_copy(bytes(effect))
    load().oalsfx_host_effect_normalize(C.byref(e))
    return e


def derive_slot(channel_format, rate, normalized_effect):
    p = desc.SlotParams()
    if not load().oalsfx_host_derive_slot(channel_format, rate, C.byref(normalized_effect), C.byref(p)):
        raise ValueError("bad channel format")
    return p


def derive_source(channel_format, rate, direct, aux, slot_types):
    n = len(slot_types)
    aux_arr = (desc.SendProps * n)(*aux)
    types = (C.c_int * n)(*slot_types)
    out = desc.SourceParams()
    if not load().oalsfx_host_derive_source(channel_format, rate, n, C.byref(direct), aux_arr, types, C.byref(out)):
        raise ValueError("bad arguments")
    return out


def preset(index):
    p = desc.ReverbProps()
    if not load().oalsfx_host_preset(index, C.byref(p)):
        raise IndexError(index)
    return load().oalsfx_host_preset_name(index).decode(), p


def preset_count():
    return load().oalsfx_host_preset_count()
