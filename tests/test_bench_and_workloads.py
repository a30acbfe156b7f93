"""Host-side pieces of the benchmark harness that need no GPU: workload builders, the CPU baseline leg, byte accounting."""
import random

import bench
from oalsfxpp_amd import desc, lib, workloads


def test_random_effect_is_deterministic_and_in_range():
    for t in range(1, 12):
        a = workloads.random_effect(random.Random(5), t)
        b = workloads.random_effect(random.Random(5), t)
        assert a.type == t and bytes(a) == bytes(b)
        n = lib.effect_normalized(a)
        m = desc.PROPS_MEMBER[t]
        # normalisation (Effect::normalize) leaves an in-range effect unchanged
        assert bytes(getattr(n.props, m)) == bytes(getattr(a.props, m)), desc.EFFECT_NAMES[t]


def test_config4_covers_every_non_null_type():
    assert sorted({workloads.config4_type(i) for i in range(22)}) == list(range(1, 12))


def test_algorithmic_bytes_match_the_survey():
    # reverb 208 B per stereo frame, chorus/flanger 32, echo 28, the rest 16; config 3's slot set 252
    bpf = workloads.BYTES_PER_FRAME
    assert bpf[desc.EAX_REVERB] == bpf[desc.REVERB] == 208 == bench.BYTES_PER_FRAME
    assert bpf[desc.CHORUS] == bpf[desc.FLANGER] == 32 and bpf[desc.ECHO] == 28
    assert all(bpf[t] == 16 for t in (desc.NULL, desc.EQUALIZER, desc.DISTORTION, desc.RING_MODULATOR, desc.COMPRESSOR,
                                      desc.DEDICATED_DIALOG, desc.DEDICATED_LFE))
    io = 16
    assert workloads.CONFIG3_BYTES_PER_FRAME == io + sum(bpf[t] - io for t in workloads.CONFIG3_CHAIN) == 252


def test_cpu_baseline_leg_reports_what_the_contract_asks():
    r = bench.cpu_baseline(target_seconds=0.3)
    from oracle import oracle as orc
    # the compiled reference is timed where its prebuilt library is present, the oracle always
    assert r["kind"] == ("reference" if orc.have_reference() else "port") and r["unit"] == "Msamples/s" and r["cores"] >= 1
    assert r["value"] > 0 and r["port_value"] > 0 and r["one_core"] > 0 and "threads" in r["sample"] and "threads" in r["port_sample"]
    if r["kind"] == "reference":
        assert "libref.so" in r["sample"]
        # the restatement does the reference's work: the two rates agree within timing noise of such a short sample
        assert 0.4 < r["value"] / r["port_value"] < 2.5
    assert 1 <= bench.usable_cores() <= 16


def test_bench_gpus_n_starts_its_own_ranks_and_never_mislabels():
    """`python bench.py --gpus 2` with no launcher: the parent (no torch, no HIP) starts two rank processes with the rendezvous
    variables set.  Without the GPUs the ranks refuse, the parent exits non-zero and no JSON line is printed -- where round 2's
    script would have benchmarked one GPU and labelled it n_gpus 1."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("covered on the GPU box by tests/test_gpu_async_and_ranks.py::test_bench_gpus_2_without_a_launcher")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    root = os.path.dirname(os.path.abspath(bench.__file__))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "2 ranks on this node but only 0 GPUs visible" in r.stderr
    # a rank whose WORLD_SIZE disagrees with --gpus refuses as well
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       env=dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
