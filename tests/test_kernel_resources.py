"""CPU: what the gfx950 code objects inside liboalsfx_hip.so say about their kernels (no GPU needed: the metadata notes of the built
library).  The cooperative reverb grids and the ring-light grid are sized for four 256-thread workgroups per CU -- 4096 instances are
exactly the 1024 workgroups the chip holds at once --, which takes at most 128 VGPRs and 40 KiB of LDS per workgroup; a build that
slips over either limit still runs, a third slower (round 3: the multichannel builds at 41 408 B of LDS, 93 instead of 66 us)."""
import os
import re
import subprocess
import tempfile

import pytest

from oalsfxpp_amd import lib

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def kernels():
    if not (os.path.exists(OBJDUMP) and os.path.exists(READELF)):
        pytest.skip("no ROCm LLVM tools here")
    out = {}
    with tempfile.TemporaryDirectory() as d:
        so = os.path.join(d, "lib.so")
        os.symlink(lib.LIB_PATH, so)
        subprocess.run([OBJDUMP, "--offloading", so], cwd=d, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(d)):
            if "gfx950" not in f:
                continue
            notes = subprocess.run([READELF, "--notes", os.path.join(d, f)], capture_output=True, text=True).stdout
            for block in notes.split("  - .agpr_count:")[1:]:
                get = lambda k: re.search(r"\." + k + r":\s+(\S+)", block).group(1)
                name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
                out[re.sub(r"\(.*", "", name).replace("void ", "").replace("oalsfx_hip::", "")] = {
                    "vgpr": int(get("vgpr_count")), "lds": int(get("group_segment_fixed_size")), "scratch": int(get("private_segment_fixed_size"))}
    return out


def test_four_workgroups_per_cu_fit():
    ks = kernels()
    # (the ring-light grid for more than two channels, k_wave_effects<8, false>, is not held to this: its bodies carry eight output channels)
    grids = {k: v for k, v in ks.items() if k.startswith(("k_reverb_steady_coop", "k_reverb_steady_kinds", "k_slot_mixed", "k_wave_effects<1,", "k_wave_effects<2,"))}
    assert len(grids) >= 30, sorted(ks)
    for name, r in grids.items():
        assert r["vgpr"] <= 128, f"{name}: {r['vgpr']} VGPRs: three workgroups per CU instead of four"
        assert r["lds"] <= 40960, f"{name}: {r['lds']} B of LDS: three workgroups per CU instead of four"


def test_the_kernels_of_a_chained_run_take_places_of_one_size():
    """A run of chained launches may alternate between two kernels (a step of two launches: the ring-light kernel, then the reverbs' grid):
    a CU hands out registers in contiguous blocks, and a 128-register wavefront does not fit the place a 120-register one gave up (measured:
    113 us per step instead of 89).  The kernels such a step launches -- the reverb builds' variants k_reverb_steady_coop_ep, the grid of
    kinds, the ring-light kernel -- allocate exactly 128 (OALSFX_EQUAL_PLACES);
    their LDS is made equal at launch time (set_lds_per_workgroup: at most 40 960 B declared, see the test above)."""
    ks = kernels()
    grids = {k: v for k, v in ks.items() if k.startswith(("k_reverb_steady_coop_ep<", "k_reverb_steady_kinds", "k_wave_effects<1,", "k_wave_effects<2,"))}
    assert len(grids) >= 20, sorted(ks)
    for name, r in grids.items():
        assert r["vgpr"] == 128, f"{name}: {r['vgpr']} registers"
        if name.startswith("k_reverb_steady_coop_ep<"):
            assert r["scratch"] == 0, f"{name}: {r['scratch']} B of scratch per lane"
    # ... and the builds a run of one kernel launches keep what they need: with every wavefront at 128 registers a full chip has none left
    # for the one-wavefront gate in front of the next launch (the headline: 41.3 us per step that way against 40.6)
    headline = ks["k_reverb_steady_coop<2, 4, false, false, false, false, false, true, false, false, false, 0>"]
    assert headline["vgpr"] <= 120, headline


def test_the_proven_builds_carry_no_scratch():
    """The FP builds have no general path inside and must not spill to memory (template arguments: channels, wavefronts, TL, HY, MD,
    ST, RG, FP, ...)."""
    for name, r in kernels().items():
        if not name.startswith("k_reverb_steady_coop<"):
            continue
        args = [a.strip() for a in name[name.index("<") + 1: name.rindex(">")].split(",")]
        if args[7] == "true":
            assert r["scratch"] == 0, f"{name}: {r['scratch']} B of scratch per lane"
