"""CPU: the oracle's sinf restatement (oracle/ref_sinf.h) against the bits recorded from the build container's libm
(tests/golden/sinf_bits.npz).  The device copy (csrc/hip/common.hpp) is exercised by the GPU parity tests through
the chorus/flanger/ring-modulator/reverb-modulation cases."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_restated_sinf_matches_recorded_libm_bits():
    gold = np.load(os.path.join(ROOT, "tests", "golden", "sinf_bits.npz"))
    src = '#include "ref_sinf.h"\nfloat f(float x) { return oracle_sinf(x); }\n'
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "s.c")
        open(c, "w").write(src)
        so = os.path.join(d, "s.so")
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "oracle"), c, "-o", so, "-lm"], check=True)
        lib = C.CDLL(so)
        lib.f.restype = C.c_float
        lib.f.argtypes = [C.c_float]
        got = np.array([lib.f(float(a)) for a in gold["args"]], dtype=np.float32).view(np.uint32)
    assert np.array_equal(got, gold["bits"])
