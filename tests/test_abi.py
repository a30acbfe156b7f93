"""CPU: the C-ABI library loads without a GPU, exports every symbol include/oalsfx_hip.h and include/oalsfx_hip_debug.h declare, agrees with
the ctypes mirror on struct sizes, and refuses loudly to create a batch when no HIP device is usable."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

from oalsfxpp_amd import desc, lib
from oalsfxpp_amd.api import Batch, BatchError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", h) for h in ("oalsfx_hip.h", "oalsfx_hip_debug.h")]


def declared_functions():
    text = "".join(open(h).read() for h in HEADERS)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(oalsfx_(?:batch|host|last|debug|device|pinned|trim|pools|group)_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 30
    so = C.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(so, n), f"{n} is declared in include/ but not exported"
        assert n in lib.SIGNATURES, f"{n} has no ctypes prototype in oalsfxpp_amd/lib.py"
    assert set(lib.SIGNATURES) == set(names)


def test_struct_sizes_match_the_c_headers():
    src = r'''
    #include <stdio.h>
    #include "oalsfx_hip_debug.h"
    int main(void) {
        printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(oalsfx_slot_params), sizeof(oalsfx_slot_state), sizeof(oalsfx_source_params),
               sizeof(oalsfx_source_state), sizeof(oalsfx_effect), sizeof(oalsfx_send_props), sizeof(oalsfx_reverb_params));
        return 0;
    }'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "s")
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        sizes = [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    expect = [C.sizeof(t) for t in (desc.SlotParams, desc.SlotState, desc.SourceParams, desc.SourceState, desc.Effect, desc.SendProps, desc.ReverbParams)]
    assert sizes == expect


def test_batch_create_fails_loudly_without_a_gpu():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(BatchError, match="No HIP device"):
        Batch(4)


def test_argument_errors_match_the_reference_messages():
    so = lib.load()
    assert not so.oalsfx_batch_create(1, 0, 48000, 1, 0)
    assert so.oalsfx_last_error() == b"Invalid channel format."
    assert not so.oalsfx_batch_create(1, desc.FMT_STEREO, 7999, 1, 0)
    assert so.oalsfx_last_error() == b"Sampling rate is out of range."
    assert not so.oalsfx_batch_create(1, desc.FMT_STEREO, 48000, 5, 0)
    assert so.oalsfx_last_error() == b"Effect count is out of range."


def test_product_never_links_the_oracle():
    out = subprocess.run(["ldd", lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out and "libref" not in out
    for dirpath, _, files in os.walk(os.path.join(ROOT, "oalsfxpp_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "liboracle" not in text and "from oracle" not in text and "import oracle" not in text, f


def test_the_pool_entry_points_answer_without_a_gpu():
    so = lib.load()
    assert so.oalsfx_pools_waiting_bytes() == 0
    assert so.oalsfx_trim_pools() == 0


def test_group_create_fails_loudly_without_a_gpu_and_names_the_device():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    so = lib.load()
    ids = (C.c_int * 2)(0, 1)
    assert not so.oalsfx_group_create(8, ids, 2, desc.FMT_STEREO, 48000, 1)
    assert so.oalsfx_group_last_error() == b"device 0: No HIP device available: the effect process path has no CPU fallback."
    assert not so.oalsfx_group_create(1, ids, 2, desc.FMT_STEREO, 48000, 1)
    assert b"fewer instances than devices" in so.oalsfx_group_last_error()
