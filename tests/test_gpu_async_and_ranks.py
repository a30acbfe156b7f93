"""GPU: the pipelined host-pointer entry point (oalsfx_batch_mix_async / oalsfx_batch_wait) against the oracle, and bench.py's real
N > 1 path rehearsed with two ranks on the one GPU (gloo between the ranks, fresh child processes)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from harness import OracleShadow, ROOT, make_effect, preset_effect, same_bits
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

E = make_effect


def test_mix_async_matches_the_oracle():
    """Api::mix semantics (reference src/oalsfxpp.cpp:3785-3829) through the pipelined entry point: many buffers in flight, a
    property change and a send change between calls, a larger call that regrows the staging slots, pageable and page-locked
    buffers, a synchronous call in between."""
    n = 12
    with Batch(n, desc.FMT_STEREO, 48000, 2) as b:
        b.set_effect(0, [preset_effect(7 * i % 113) for i in range(n)])
        b.set_effect_type(1, desc.CHORUS)
        b.apply_changes()
        shadows = [OracleShadow(b, i) for i in range(n)]
        script = [256] * 9 + ["change"] + [256] * 4 + [1024, 100, 256, "sync-call", 256, 256]
        pending = []  # (input, output array) of the calls in flight

        def collect():
            b.wait()
            for x, y in pending:
                for i, s in enumerate(shadows):
                    ok, nbad = same_bits(y[i], s.mix(x[i]))
                    assert ok, f"instance {i}: {nbad} samples differ"
            pending.clear()

        k = 0
        for op in script:
            if op == "change":
                collect()   # the shadows read the descriptors back: they must see the old ones for the calls in flight
                b.set_effect(0, preset_effect(40), first=3, count=2)
                b.set_send_props(-1, 0.8, 0.5, 1.0, first=5, count=1)
                b.apply_changes()
                for s in shadows:
                    s.sync()
                continue
            if op == "sync-call":
                collect()
                x = np.stack([orc.synth(40 + i, k, 512).reshape(256, 2) for i in range(n)])
                y = b.mix(x)
                for i, s in enumerate(shadows):
                    assert same_bits(y[i], s.mix(x[i]))[0]
                k += 1
                continue
            frames = op
            pinned = k % 3 != 2
            x = b.pinned_array(frames) if pinned else np.empty((n, frames, 2), dtype=np.float32)
            y = b.pinned_array(frames) if pinned else np.empty((n, frames, 2), dtype=np.float32)
            x[:] = np.stack([orc.synth(40 + i, k, frames * 2).reshape(frames, 2) for i in range(n)])
            b.mix_async(x, y)
            pending.append((x, y))
            k += 1
        collect()
        for i, s in enumerate(shadows):
            d = s.compare_state()
            assert not d, f"instance {i}: " + "; ".join(d[:4])


def test_api_facade_places_instances_with_oalsfx_device(tmp_path):
    """OALSFX_DEVICE picks the HIP ordinal for oalsfxpp::Api::initialize; an ordinal that does not exist fails with a message."""
    exe, out = str(tmp_path / "dropin"), str(tmp_path / "out.f32")
    from oalsfxpp_amd import lib
    libdir = os.path.dirname(lib.LIB_PATH)
    subprocess.run(["g++", "-std=c++14", "-O1", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "api_dropin.cpp"),
                    "-L", libdir, "-loalsfx_hip", f"-Wl,-rpath,{libdir}", "-o", exe], check=True)
    ok = subprocess.run([exe, out], capture_output=True, text=True, env=dict(os.environ, OALSFX_DEVICE="0"))
    assert ok.returncode == 0, ok.stderr + ok.stdout
    bad = subprocess.run([exe, out], capture_output=True, text=True, env=dict(os.environ, OALSFX_DEVICE="63"))
    assert bad.returncode != 0 and "ordinal" in (bad.stderr + bad.stdout)
    bad = subprocess.run([exe, out], capture_output=True, text=True, env=dict(os.environ, OALSFX_DEVICE="gpu1"))
    assert bad.returncode != 0 and "ordinal" in (bad.stderr + bad.stdout)


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """`bench.py --gpus 2` as the driver launches it -- one process per rank, rendezvous, device pick, barriers around the timed
    region, max over ranks, rank 0 printing the line -- with gloo between the ranks because both share the one GPU here.  The
    children are fresh processes started before this test's own GPU work matters to them (nothing is exec'ed from a process that
    has touched the GPU: subprocess forks and execs a new interpreter, which initialises HIP itself)."""
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OALSFX_DIST_BACKEND="gloo", OALSFX_DUMP_OUTPUT=str(tmp_path / f"rank{rank}_dst.npy"))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "8",
                                       "--no-cpu-baseline", "--instances", "64", "--spin-up-ms", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-2000:]
    line = [l for l in outs[0][0].splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "batch-split x2, no collectives" and d["value"] > 0
    assert d["scaling"] == "weak" and d["steps"] == 8 and d["warmup"] == 8
    assert [r["rank"] for r in d["config"]["devices"]] == [0, 1] and all(r["pci_bus_id"] for r in d["config"]["devices"])
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")], "only rank 0 prints the line"
    assert d["config5"]["instances_total"] == 2 * 32768 and d["config5"]["value"] > 0
    # both ranks' last outputs against a single-process run of the same global instance ranges
    from oalsfxpp_amd import workloads
    got = [np.load(str(tmp_path / f"rank{rank}_dst.npy")) for rank in range(2)]
    import torch
    with Batch(128, desc.FMT_STEREO, 48000, 1) as b:
        workloads.setup(b, "config2")
        src = [torch.empty(128 * 512, dtype=torch.float32, device="cuda") for _ in range(8)]
        dst = torch.empty(128 * 512, dtype=torch.float32, device="cuda")
        # rank r's inputs: buffer k of its ring is fill_synthetic(k + 1000 r) over its 64 instances (instance index local to the rank)
        for r in range(2):
            with Batch(64, desc.FMT_STEREO, 48000, 1) as tmp:
                for k in range(8):
                    part = torch.empty(64 * 512, dtype=torch.float32, device="cuda")
                    tmp.fill_synthetic(256, k + 1000 * r, part.data_ptr())
                    tmp.synchronize()
                    src[k][r * 64 * 512:(r + 1) * 64 * 512] = part
        torch.cuda.synchronize()
        # warm-up, timed region, kernel statistics (bench.KERNEL_STATS_STEPS), the fixed roofline region (bench.ROOFLINE_WARMUP +
        # ROOFLINE_LAUNCHES), the steady state of chained launches (ROOFLINE_WARMUP + CHAINED_RUN; 64 instances are whole workgroups).  No
        # device spin-up (--spin-up-ms 0): it runs for a time, not for a number of steps.
        steps = 8 + 8 + 32 + 64 + 128 + 64 + 512
        for k in range(steps):
            b.mix_device(256, src[k % 8].data_ptr(), dst.data_ptr())
        b.synchronize()
        want = dst.cpu().numpy()
    for r in range(2):
        assert got[r].tobytes() == want[r * 64 * 512:(r + 1) * 64 * 512].tobytes(), f"rank {r}: outputs differ from the single-process run"


def test_bench_gpus_2_without_a_launcher(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: the script starts its own two rank processes (before it touches torch
    or HIP) and relays rank 0's line; it never benchmarks one GPU under a --gpus 2 label.  gloo between the ranks because both share
    the one GPU here; with the default backend the same command must refuse instead."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "6", "--no-cpu-baseline", "--instances", "64", "--spin-up-ms", "0",
           "--no-config5"]
    r = subprocess.run(cmd, env=dict(env, OALSFX_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "one line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and [x["rank"] for x in d["config"]["devices"]] == [0, 1] and d["value"] > 0
    assert d["config"]["parallelism"] == "batch-split x2, no collectives"
    # one visible GPU, RCCL between the ranks, no per-rank visibility mask: refused, not mislabelled
    import torch
    if torch.cuda.device_count() == 1:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert "2 ranks on this node but only 1 GPUs visible" in r.stderr
