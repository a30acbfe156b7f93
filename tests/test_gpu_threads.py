"""GPU: callers on several host threads.  The reference's contract ("Threading"): one Api object is not thread-safe,
distinct objects share nothing mutable and may be driven in parallel.  Here: two host threads, each with its own Batch (one
reverb-heavy, one of ring-light effects), mixing, changing properties and reading state at the same time, each bit-exact against
its own oracle shadows; and eight oalsfxpp::Api objects through the C++ facade, threaded against serial."""
import os
import subprocess
import threading

import numpy as np
import pytest

from harness import OracleShadow, ROOT, make_effect, preset_effect, same_bits
from oalsfxpp_amd import desc, lib
from oalsfxpp_amd.api import Batch
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _drive(kind, start, errors):
    """One caller thread: its own batch, its own shadows, 14 calls with changes in between."""
    try:
        import random
        from oalsfxpp_amd.workloads import random_effect
        n = 96
        ring_light = [desc.CHORUS, desc.FLANGER, desc.ECHO, desc.EQUALIZER, desc.DISTORTION, desc.RING_MODULATOR, desc.COMPRESSOR, desc.DEDICATED_DIALOG]
        with Batch(n, desc.FMT_STEREO, 48000, 2 if kind == "reverbs" else 1) as b:
            if kind == "reverbs":
                b.set_effect(0, [preset_effect((5 * i) % 113) for i in range(n)])
                b.set_effect_type(1, desc.ECHO)
            else:
                b.set_effect(0, [random_effect(random.Random(900 + i), ring_light[i % len(ring_light)]) for i in range(n)])
            b.apply_changes()
            sample = list(range(0, n, 7)) + [n - 1]
            shadows = {i: OracleShadow(b, i) for i in sample}
            start.wait()
            for k, frames in enumerate([256, 256, 256, 256, 100, 256, 256, 441, 256, 256, 2048, 256, 256, 256]):
                if k == 5:
                    if kind == "reverbs":
                        b.set_effect(0, preset_effect(40), first=7, count=8)     # cross-fades beside steady neighbours
                        b.set_send_props(-1, 0.8, 0.6, 1.0, first=14, count=2)   # a send filter
                    else:
                        b.set_effect(0, make_effect(desc.EAX_REVERB), first=21, count=3)  # a type change: state and rings re-created
                    b.apply_changes()
                x = np.stack([orc.synth((7000 if kind == "reverbs" else 8000) + i, k, frames * 2).reshape(frames, 2) for i in range(n)])
                y = b.mix(x)
                for i in sample:
                    ok, nbad = same_bits(y[i], shadows[i].mix(x[i]))
                    assert ok, f"{kind}: instance {i}, call {k}: {nbad} samples differ"
            for i in sample:
                d = shadows[i].compare_state()
                assert not d, f"{kind}: instance {i}: " + "; ".join(d[:4])
    except BaseException as e:  # noqa: BLE001 (reported by the test's own thread)
        errors.append(f"{kind}: {e!r}")


def test_two_batches_on_two_host_threads():
    errors = []
    start = threading.Barrier(2)
    threads = [threading.Thread(target=_drive, args=(kind, start, errors)) for kind in ("reverbs", "ring-light")]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, "; ".join(errors)


def test_api_objects_on_four_threads_match_the_serial_run(tmp_path):
    """tests/cpp/api_threads.cpp: eight oalsfxpp::Api objects (different effects, formats, rates, call sizes, a change while
    streaming), run one after the other and then on four threads at once: every output bit-identical."""
    exe = str(tmp_path / "api_threads")
    libdir = os.path.dirname(lib.LIB_PATH)
    subprocess.run(["g++", "-std=c++14", "-O1", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "api_threads.cpp"),
                    "-L", libdir, "-loalsfx_hip", f"-Wl,-rpath,{libdir}", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout


def test_a_group_of_two_shards_on_one_gpu_matches_one_batch(tmp_path):
    """tests/cpp/group_two_shards.cpp: the multi-GPU split of the C ABI (oalsfx_group_*: a contiguous instance range, a batch and a host
    thread per device, no collective), rehearsed with both shards on device 0 -- every buffer bit-identical to one batch of all the
    instances, setters that straddle the shard boundary, a change while streaming."""
    exe = str(tmp_path / "group_two_shards")
    libdir = os.path.dirname(lib.LIB_PATH)
    subprocess.run(["g++", "-std=c++14", "-O1", "-pthread", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "group_two_shards.cpp"),
                    "-L", libdir, "-loalsfx_hip", f"-Wl,-rpath,{libdir}", "-o", exe], check=True)
    for n in ("37", "2048"):
        r = subprocess.run([exe, n], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, r.stderr + r.stdout


def test_group_device_buffers_and_three_shards():
    """The device-buffer entry point of a group (queued shard after shard, consecutive calls chained per shard) against single batches of
    the same instance ranges; three shards of unequal size on device 0."""
    import torch
    from oalsfxpp_amd.api import Group
    n, frames, calls = 70, 256, 12
    with Group(n, [0, 0, 0], desc.FMT_STEREO, 48000, 1) as g:
        assert [c for _, _, c in g.shards] == [24, 23, 23] and [f for _, f, _ in g.shards] == [0, 24, 47]
        effects = [preset_effect((3 * i) % 113) for i in range(n)]
        g.set_effect(0, effects)
        g.apply_changes()
        xs = [np.stack([orc.synth(900 + i, k, frames * 2).reshape(frames, 2) for i in range(n)]) for k in range(calls)]
        outs = []
        for _, f, c in g.shards:
            dx = [torch.from_numpy(x[f:f + c].copy()).cuda() for x in xs]
            outs.append((dx, [torch.empty_like(d) for d in dx]))
        torch.cuda.synchronize()
        for k in range(calls):
            g.mix_device(frames, [o[0][k].data_ptr() for o in outs], [o[1][k].data_ptr() for o in outs])
        g.synchronize()
        with Batch(n, desc.FMT_STEREO, 48000, 1) as b:
            b.set_effect(0, effects)
            b.apply_changes()
            for k in range(calls):
                want = b.mix(xs[k])
                got = np.concatenate([o[1][k].cpu().numpy() for o in outs])
                ok, nbad = same_bits(got, want)
                assert ok, f"call {k}: {nbad} samples differ"


def test_api_array_matches_separate_api_objects(tmp_path):
    """tests/cpp/api_array.cpp: oalsfxpp::ApiArray (include/oalsfxpp_array.h) -- the reference's Api surface for many chains at once, each
    with buffers of its own (oalsfx_batch_mix_gather) -- against the same forty chains as forty oalsfxpp::Api objects: bit-identical."""
    exe = str(tmp_path / "api_array")
    libdir = os.path.dirname(lib.LIB_PATH)
    subprocess.run(["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "api_array.cpp"),
                    "-L", libdir, "-loalsfx_hip", f"-Wl,-rpath,{libdir}", "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr + r.stdout
