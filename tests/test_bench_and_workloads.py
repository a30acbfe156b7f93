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
