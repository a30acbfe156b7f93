#!/usr/bin/env python3
"""Headline benchmark: EAX reverb, 256-frame stereo buffers, 4096 independent instances per GPU.

    python bench.py --gpus 1 --steps 200 --warmup 64
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one `oalsfx_batch_mix_device` call = every instance advanced by one 256-frame buffer
(BASELINE.json configs[1]; per-GPU work is fixed as N grows -> weak scaling, no collective on the data
path: instances are independent).  Inputs (synthetic uniform noise) are generated in
device memory before the timed region.  Rank 0 prints one JSON line.

Extra objects in the line:
  roofline      algorithmic bytes (208 B per stereo frame) / live HIP-event duration of the
                reverb kernel on its launch stream, against the 8 TB/s HBM3E peak
  cpu_baseline  the compiled reference (oracle/_ref/libref.so, kind "reference": built in the build container from
                /root/reference and carried along prebuilt) or, where that file is absent, the CPU oracle
                (oracle/liboracle.so, kind "port"), timed on this host's cores on a bounded sample of the same
                workload (rank 0, N=1 only); the oracle's own figure is always reported beside it as port_value
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FRAMES = 256
CHANNELS = 2
BYTES_PER_FRAME = 208          # 16 B I/O + 24 fp32 delay-line reads + 24 fp32 delay-line writes
TIMED_EVERY = int(os.environ.get("OALSFX_TIMED_EVERY", "8"))   # (the config5 leg: every 8th step carries events)
KERNEL_STATS_STEPS = 32        # launches timed one by one for the `kernels` object, outside the timed region
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
VALU_SIMDS = 1024              # 256 CUs x 4 SIMDs
VALU_CLOCK_GHZ = 2.4
VALU_CYCLES_PER_WAVE_INSTRUCTION = 4   # a wave64 VALU instruction holds its 16-lane SIMD for four cycles
METRIC = "Msamples/sec EAX reverb, 256-frame buffers, batch=4096; % HBM roofline"
METRICS = {
    "config3": "Msamples/sec 4-slot chain (chorus, flanger, echo, EAX reverb), 256-frame buffers, batch=4096; % HBM roofline",
    "config4": "Msamples/sec 11 effect types mixed, randomised properties, 256-frame buffers, batch=8192; % HBM roofline",
}
WORKLOADS = {
    "config2": "{n} independent EAX-reverb instances per GPU, stereo, 48 kHz, 256-frame buffers, 1 slot, default properties "
               "(BASELINE.json configs[1])",
    "config2-presets": "{n} independent EAX-reverb instances per GPU, stereo, 48 kHz, 256-frame buffers, 1 slot, "
                       "EFX preset i%113 per instance (BASELINE.json configs[1], robustness run)",
    "config3": "{n} instances per GPU x 4 parallel slots (chorus, flanger, echo, EAX reverb; defaults), stereo, 48 kHz, "
               "256-frame buffers (BASELINE.json configs[2])",
    "config4": "{n} instances per GPU, 1 slot, effect type 1 + i%11, every property uniform in its range (seed = instance), "
               "stereo, 48 kHz, 256-frame buffers (BASELINE.json configs[3])",
    "config5": "{n} independent EAX-reverb instances per GPU (one GPU's share of BASELINE.json configs[4]: 262144 over 8 GPUs), stereo, "
               "48 kHz, 256-frame buffers, 1 slot, default properties",
}


def usable_cores():
    """Cores this process may really use: affinity mask, capped by a cgroup CPU quota when there is one."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(cores, int(os.environ.get("OALSFX_CPU_THREADS", "16")))  # the GPU box gives 16 cores per GPU


def cpu_baseline(target_seconds=12.0):
    """Times the CPU oracle on a bounded sample: default-parameter EAX reverb, stereo, 256-frame buffers."""
    import ctypes as C
    from oalsfxpp_amd import desc, lib
    from oracle import oracle as orc

    threads = usable_cores()
    e = lib.effect_normalized(lib.effect_defaults(desc.EAX_REVERB))
    p = lib.derive_slot(desc.FMT_STEREO, 48000, e)
    p.update_seq = 1
    sp = lib.derive_source(desc.FMT_STEREO, 48000, desc.SendProps(1, 1, 1), [desc.SendProps(1, 1, 1)], [desc.EAX_REVERB])
    o = orc.Oracle(CHANNELS, 1)
    o.set_source(sp)
    o.set_slot(0, p, restart=True)
    instances = 4 * threads
    warm = 8
    # calibrate with a short run, then size the timed run for ~target_seconds
    t = o.bench(instances, FRAMES, warm, 64, threads)
    rate = instances * 64 * FRAMES / t
    buffers = max(32, int(target_seconds * rate / (instances * FRAMES)))
    t = o.bench(instances, FRAMES, warm, buffers, threads)
    # one core, for scale: a short sample of the same workload
    b1 = max(32, buffers // 8)
    t1 = o.bench(4, FRAMES, warm, b1, 1)
    # the same source at -O3 -march=x86-64-v3 with FMA contraction allowed (within 1e-5 of the parity build,
    # tests/test_oracle_fast.py): the fastest fair scalar CPU figure
    fast_value = None
    try:
        of = orc.Oracle(CHANNELS, 1, fast=True)
        of.set_source(sp)
        of.set_slot(0, p, restart=True)
        bf = max(32, buffers // 2)
        tf = of.bench(instances, FRAMES, warm, bf, threads)
        fast_value = round(instances * bf * FRAMES / tf / 1e6, 3)
    except (FileNotFoundError, OSError):
        pass
    # the compiled reference itself (oracle/_ref/libref.so, built in the build container and carried along prebuilt), when present
    ref_value = ref_sample = None
    if orc.have_reference() and hasattr(orc.ref_lib(), "ref_bench"):
        ref_bench = orc.ref_lib().ref_bench
        tr = ref_bench(desc.FMT_STEREO, 48000, C.byref(e), instances, FRAMES, warm, 64, threads)  # calibration
        br = max(32, int(0.6 * target_seconds * 64 / tr)) if tr > 0 else 32
        tr = ref_bench(desc.FMT_STEREO, 48000, C.byref(e), instances, FRAMES, warm, br, threads)
        if tr > 0:
            ref_value = round(instances * br * FRAMES / tr / 1e6, 3)
            ref_sample = (f"{instances} reference Api objects (EAX reverb, stereo, 48 kHz) x {br} Api::mix calls of {FRAMES} frames after {warm} "
                          f"warm-up calls, {threads} threads, oracle/_ref/libref.so (g++ -O2 -ffp-contract=off), {tr:.1f} s")
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    port_value = round(instances * buffers * FRAMES / t / 1e6, 3)
    port_sample = (f"{instances} EAX-reverb instances x {buffers} buffers of {FRAMES} stereo frames after {warm} warm-up buffers, "
                   f"{threads} threads, oracle/liboracle.so (-O2 -ffp-contract=off), {t:.1f} s")
    return {
        "value": ref_value if ref_value is not None else port_value,
        "unit": "Msamples/s",
        "cores": threads,
        "kind": "reference" if ref_value is not None else "port",
        "sample": ref_sample if ref_value is not None else port_sample,
        "port_value": port_value,
        "port_sample": port_sample,
        "one_core": round(4 * b1 * FRAMES / t1 / 1e6, 3),
        "value_o3_x86_64_v3_fma": fast_value,
        "cpu": f"{model}, {os.cpu_count()} logical CPUs on the host",
    }


def percentile(sorted_values, q):
    if not sorted_values:
        return 0.0
    k = (len(sorted_values) - 1) * q
    lo = int(k)
    hi = min(lo + 1, len(sorted_values) - 1)
    return sorted_values[lo] + (sorted_values[hi] - sorted_values[lo]) * (k - lo)


ROOFLINE_WARMUP = 64       # the fixed internal region the roofline object is measured on, whatever --steps / --warmup say
ROOFLINE_LAUNCHES = 128
CHAINED_RUN = 512          # launches of the steady-state measurement of chained launches (roofline.chained_launches)


def device_record(rank, ordinal):
    """Which GPU this rank runs on: HIP ordinal inside the process's visibility mask, PCI bus id, marketing name."""
    import ctypes as C
    import torch
    from oalsfxpp_amd import lib
    buf = C.create_string_buffer(64)
    so = lib.load()
    pci = buf.value.decode() if hasattr(so, "oalsfx_device_pci_bus_id") and so.oalsfx_device_pci_bus_id(ordinal, buf, 64) else None
    return {"rank": rank, "ordinal": ordinal, "pci_bus_id": pci or buf.value.decode() or None, "name": torch.cuda.get_device_name(ordinal),
            "visible_devices": os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")}


def timed_region(batch, src, dst, n_in, first_step, steps, sharding, backend):
    """`steps` mix calls bracketed by barrier + synchronize on both sides; returns the MAX over ranks of the wall time."""
    import torch
    batch.synchronize()
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        batch.mix_device(FRAMES, src[(first_step + k) % n_in].data_ptr(), dst.data_ptr())
    batch.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sharding.barrier()
    torch.cuda.synchronize()
    return sharding.max_over_ranks(elapsed, device="cuda" if backend == "nccl" else "cpu")


def resident_inputs(batch, n, rank, n_in=8):
    """Inputs resident in HBM: a ring of pre-generated buffers, one output buffer."""
    import torch
    floats = n * FRAMES * CHANNELS
    src = [torch.empty(floats, dtype=torch.float32, device="cuda") for _ in range(n_in)]
    dst = torch.empty(floats, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for k, s in enumerate(src):
        batch.fill_synthetic(FRAMES, k + 1000 * rank, s.data_ptr())
    batch.synchronize()
    return src, dst


def roofline_region(batch, src, dst, n_in, first_step, bytes_per_launch):
    """The dominant kernel alone, on a region of fixed length: ROOFLINE_WARMUP untimed launches, then ROOFLINE_LAUNCHES launches
    each bracketed by HIP events on the launch stream; the empty-pair reading is taken off every sample; frac uses the median."""
    from oalsfxpp_amd import desc
    for k in range(ROOFLINE_WARMUP):
        batch.mix_device(FRAMES, src[(first_step + k) % n_in].data_ptr(), dst.data_ptr())
    batch.synchronize()
    batch.kernel_timing(1)
    for k in range(ROOFLINE_LAUNCHES):
        batch.mix_device(FRAMES, src[(first_step + ROOFLINE_WARMUP + k) % n_in].data_ptr(), dst.data_ptr())
    batch.synchronize()
    raw = sorted(batch.kernel_timing_samples(desc.EAX_REVERB))
    general = batch.kernel_timing_read(desc.REVERB + 16)[0]
    batch.kernel_timing(0)
    bracket_us = batch.event_overhead(200)
    us = [max(x - bracket_us, 0.0) for x in raw]
    med = percentile(us, 0.5)
    achieved = bytes_per_launch / (med * 1e-6) / 1e9 if med > 0 else 0.0
    return {
        "bound": "hbm",
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": None,
        "kernel": batch.last_reverb_kernel,
        "kernel_us": round(med, 2),
        "kernel_us_min": round(us[0], 2) if us else None,
        "kernel_us_p10": round(percentile(us, 0.1), 2),
        "kernel_us_p90": round(percentile(us, 0.9), 2),
        "kernel_us_max": round(us[-1], 2) if us else None,
        "kernel_us_mean": round(sum(us) / max(len(us), 1), 2),
        "kernel_us_event_pair": round(percentile(raw, 0.5), 2),
        "event_pair_empty_us": round(bracket_us, 2),
        "launches_timed": len(us),
        "region": f"{ROOFLINE_WARMUP} untimed + {ROOFLINE_LAUNCHES} event-timed launches after the CLI-timed region; frac from the median",
        "general_kernel_launches_in_region": general,
        "algorithmic_bytes_per_launch": bytes_per_launch,
    }


def spawn_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher: this process (which has neither imported torch nor touched HIP) starts N fresh
    rank processes of the same command line, one per GPU, with the rendezvous variables torch.distributed.run would set; rank 0's
    output is relayed.  Returns the exit status: non-zero if any rank failed."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for rank in range(n_ranks):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OALSFX_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if rank == 0 else subprocess.DEVNULL))
    status = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            rc = p.poll()
            if rc is None:
                continue
            pending.remove(p)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                for q in pending:  # a rank that failed would leave the others waiting at the rendezvous
                    q.terminate()
        time.sleep(0.05)
    return status


def in_process_group(args):
    """`--in-process`: the same headline loop through the C ABI's group object (oalsfx_group_*): ONE process, a batch and a
    host thread per device, device-resident buffers per shard, no collective.  What a C++ caller of configs[4] gets without a launcher;
    the driver's contract (one process per GPU, RCCL barrier, max over ranks) is the default mode above, and both can be run on an
    8-GPU node.  `--devices 0,0` rehearses two shards on one GPU."""
    import torch
    from oalsfxpp_amd import desc
    from oalsfxpp_amd.api import Group
    devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    n = args.instances or 4096
    g = Group(n * len(devices), devices, desc.FMT_STEREO, 48000, 1)
    g.set_effect_type(0, desc.EAX_REVERB)
    g.apply_changes()
    n_in = 8
    bufs = []
    for d, first, count in g.shards:
        with torch.cuda.device(d):
            src = [torch.empty(count * FRAMES * 2, dtype=torch.float32, device=f"cuda:{d}").uniform_(-1.0, 1.0) for _ in range(n_in)]
            bufs.append((src, torch.empty_like(src[0])))
    for d in set(devices):
        torch.cuda.synchronize(d)

    def step(k):
        g.mix_device(FRAMES, [b[0][k % n_in].data_ptr() for b in bufs], [b[1].data_ptr() for b in bufs])

    t0 = time.perf_counter()
    k = 0
    while time.perf_counter() - t0 < max(args.spin_up_ms, 1.0) * 1e-3 or k < 8:
        step(k); k += 1
        if k <= 4 or k % 64 == 0:
            g.synchronize()
    for k in range(args.warmup):
        step(k)
    g.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    g.synchronize()
    elapsed = time.perf_counter() - t0
    total_frames = n * len(devices) * FRAMES * args.steps
    step_gbs = BYTES_PER_FRAME * n * len(devices) * FRAMES / (elapsed / args.steps) / 1e9
    print(json.dumps({
        "metric": METRIC, "value": round(total_frames / elapsed / 1e6, 3), "unit": "Msamples/s", "n_gpus": len(set(devices)), "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": WORKLOADS["config2"].format(n=n), "instances_per_gpu": n, "frames_per_buffer": FRAMES,
                   "parallelism": f"one process, oalsfx_group over {len(devices)} shard(s) on devices {devices}: a batch and a host thread per shard, no collectives",
                   "shards": [{"device": d, "first": f, "count": c} for d, f, c in g.shards]},
        "roofline": {"bound": "hbm", "achieved": round(step_gbs, 1), "peak": HBM_PEAK_GBS * len(set(devices)), "unit": "GB/s",
                     "frac": round(step_gbs / (HBM_PEAK_GBS * len(set(devices))), 4), "traffic": None,
                     "kernel": "whole step over all shards (algorithmic bytes of the step / step time); the kernel alone: the default mode"},
    }), flush=True)
    g.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--instances", type=int, default=0, help="instances per GPU (default 4096; 8192 for config4, 32768 for config5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="experiment: no HIP events around the launches (roofline fields then use the step time)")
    ap.add_argument("--preset-mix", action="store_true", help="robustness run: instance i uses EFX preset i %% 113")
    ap.add_argument("--spin-up-ms", type=float, default=60.0, help="untimed steps of the workload for this long before the warm-up steps, so "
                    "that the card has reached its clocks when the short timed region starts (0: off)")
    ap.add_argument("--no-chain", action="store_true", help="consecutive calls in plain stream order (no overlap of a launch's tail with the "
                    "next one's head): the configuration whose rocprofv3 kernel durations are those of the kernel alone")
    ap.add_argument("--preset", type=int, default=-1, help="experiment: every instance uses EFX preset N")
    ap.add_argument("--workload", default="config2", choices=["config2", "config3", "config4", "config5"],
                    help="BASELINE.json configs[1] (default, the headline metric), configs[2] (4-slot chain), configs[3] (11 effect types, "
                         "randomised properties; 8192 instances unless --instances is given) or one GPU's share of configs[4] "
                         "(32768 EAX reverbs per GPU)")
    ap.add_argument("--config5", action="store_true", help="also time one GPU's share of configs[4] (32768 instances per GPU) and report it as "
                                                           "the `config5` object; always on when --gpus > 1")
    ap.add_argument("--host-io", type=int, default=10, metavar="K",
                    help="after the timed region, also time K steps through the host-pointer entry points (PCIe both ways)")
    ap.add_argument("--no-other-configs", action="store_true", help="leave out the `other_configs` object (configs[2] and configs[3], 100 steps each, "
                                                                    "in single-GPU runs of the default workload)")
    ap.add_argument("--no-config5", action="store_true", help="multi-GPU runs: leave the `config5` object out (rehearsals on one GPU)")
    ap.add_argument("--in-process", action="store_true", help="time the C ABI's group object instead (oalsfx_group_*: one process, a batch and a host "
                                                              "thread per device); the driver's contract is the default, one process per GPU")
    ap.add_argument("--devices", default="", help="--in-process: comma-separated HIP ordinals of the shards (default 0 .. gpus - 1; 0,0 rehearses two "
                                                  "shards on one GPU)")
    args = ap.parse_args()

    if args.in_process:
        if args.no_chain:
            os.environ["OALSFX_DEBUG_FLAGS"] = hex(int(os.environ.get("OALSFX_DEBUG_FLAGS", "0"), 0) | 0x400)
        return in_process_group(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: be the launcher (before torch or HIP are touched in this process)
        sys.exit(spawn_ranks(args.gpus))

    if args.no_chain:
        # (read once by the library, at its first call: before it is loaded)
        os.environ["OALSFX_DEBUG_FLAGS"] = hex(int(os.environ.get("OALSFX_DEBUG_FLAGS", "0"), 0) | 0x400)

    import torch
    import torch.distributed as dist

    from oalsfxpp_amd import desc, lib, sharding, workloads
    from oalsfxpp_amd.api import Batch

    rank, world, local_rank = sharding.env_rank_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would misstate n_gpus")
    distributed = world > 1
    # RCCL ("nccl") between the ranks; OALSFX_DIST_BACKEND=gloo is for rehearsing the N > 1 path where the ranks share a GPU
    backend = os.environ.get("OALSFX_DIST_BACKEND", "nccl")
    # one process per GPU: LOCAL_RANK picks the device; if the launcher narrowed the visible devices to one per process
    # (HIP_VISIBLE_DEVICES), that one is device 0
    visible = torch.cuda.device_count()
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    # (a launcher that narrows every rank's visibility to its own GPU leaves one visible device per rank: trusted; this script's own
    # ranks all inherit the same mask)
    if distributed and backend == "nccl" and visible < local_world and (visible != 1 or os.environ.get("OALSFX_BENCH_SELF_LAUNCHED")):
        raise SystemExit(f"--gpus {args.gpus}: {local_world} ranks on this node but only {visible} GPUs visible (one process per GPU; "
                         "OALSFX_DIST_BACKEND=gloo rehearses ranks that share a GPU on purpose)")
    local_rank = local_rank % visible if visible > 0 else local_rank
    torch.cuda.set_device(local_rank)
    if distributed:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    mine = device_record(rank, local_rank)
    devices = [mine]
    if distributed:
        devices = [None] * world
        dist.all_gather_object(devices, mine)

    config5_main = args.workload == "config5"
    n = args.instances or {"config4": 8192, "config5": 32768}.get(args.workload, 4096)
    workload = "config2-presets" if args.preset_mix else ("config2" if config5_main else args.workload)
    batch = Batch(n, desc.FMT_STEREO, 48000, workloads.effect_count(workload), device_id=local_rank)
    if args.preset >= 0:
        e = lib.effect_defaults(desc.EAX_REVERB)
        name, e.props.reverb = lib.preset(args.preset)
        batch.set_effect(0, e)
        batch.apply_changes()
        workload = "config2-presets"
        WORKLOADS[workload] = "{n} EAX-reverb instances per GPU, all EFX preset " + name
    else:
        workloads.setup(batch, workload, first_instance=rank * n)

    n_in = 8
    src, dst = resident_inputs(batch, n, rank, n_in)
    # The card takes tens of milliseconds of continuous load to reach its clocks (the same loop runs 45 us per step for its first 16 ms and
    # 39 afterwards, chained or not: scripts/chain_probe.py), and --steps 20 --warmup 5 is one millisecond: the device is brought up to
    # speed first, with the workload's own steps, untimed, before the W warm-up steps the command line asks for.  (--spin-up-ms 0: off.)
    if args.spin_up_ms > 0:
        for k in range(2):
            batch.mix_device(FRAMES, src[k % n_in].data_ptr(), dst.data_ptr())
            batch.synchronize()
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < args.spin_up_ms * 1e-3:
            for k in range(64):
                batch.mix_device(FRAMES, src[k % n_in].data_ptr(), dst.data_ptr())
            batch.synchronize()
    for k in range(args.warmup):
        batch.mix_device(FRAMES, src[k % n_in].data_ptr(), dst.data_ptr())
        if k < 2:
            # fresh reverbs cross-fade through their first buffer; once the device has reported them settled (the read-back is looked at
            # by a synchronising call) they are listed for the steady-state builds: let that happen inside the warm-up, however short
            batch.synchronize()

    # the timed region carries no events: an event pair around a launch costs that step about 10 us of dispatch overhead (the roofline
    # region and the short region below have them instead)
    batch.synchronize()
    batch.kernel_timing(0)
    chained_before = batch.chained_calls
    elapsed = timed_region(batch, src, dst, n_in, args.warmup, args.steps, sharding, backend)
    chained = batch.chained_calls - chained_before

    # per effect type: launches and summed HIP-event duration of its kernel(s), over a short region of its own behind the timed one
    if not args.no_kernel_timing:
        batch.kernel_timing(1)
        for k in range(KERNEL_STATS_STEPS):
            batch.mix_device(FRAMES, src[(args.warmup + args.steps + k) % n_in].data_ptr(), dst.data_ptr())
        batch.synchronize()
    kernels = {}
    timed = {"wave_effects (all ring-light types of a slot, one launch)": desc.CHORUS, "reverb + eax_reverb steady-state": desc.EAX_REVERB,
             "reverb + eax_reverb general": desc.REVERB + 16,
             "slot_mixed (ring-light types + steady reverbs of a slot, one grid)": 32}
    for name, t in timed.items():
        l, ms = batch.kernel_timing_read(t)
        if l:
            kernels[name] = {"launches": l, "avg_us": round(ms / l * 1e3, 2)}
    batch.kernel_timing(False)
    frames_per_launch = n * FRAMES
    if workload == "config3":
        bytes_per_step = workloads.CONFIG3_BYTES_PER_FRAME * frames_per_launch
    elif workload == "config4":
        bytes_per_step = sum(workloads.BYTES_PER_FRAME[workloads.config4_type(rank * n + i)] for i in range(n)) * FRAMES
    else:
        bytes_per_step = BYTES_PER_FRAME * frames_per_launch
    plan = batch.plan(workloads.effect_count(workload) - 1)
    if workload == "config2" and not args.no_kernel_timing:
        # the headline: the dominant kernel alone, on its own fixed region
        roofline = roofline_region(batch, src, dst, n_in, args.warmup + args.steps + KERNEL_STATS_STEPS, bytes_per_step)
    else:
        # several kernels share a step, some of them side by side on forked streams: the step's algorithmic bytes
        # against the step time itself
        step_s = elapsed / args.steps
        achieved = bytes_per_step / step_s / 1e9
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                    "traffic": None, "kernel": "all effect kernels of a step (algorithmic bytes of the step / step time)",
                    "kernel_us": round(step_s * 1e6, 2), "launches_timed": args.steps, "algorithmic_bytes_per_launch": bytes_per_step}
    issue_file = os.path.join(ROOT, "profiles", "valu_issue.json")
    if workload in ("config3", "config4") and os.path.exists(issue_file) and n == {"config3": 4096, "config4": 8192}[workload]:
        # These steps are not bound by bytes (16 to 32 B per frame for most of their effect types) but by instruction issue: serial filter
        # recurrences on a few lanes still hold a SIMD for four cycles per wave instruction.  The bound that fits: the share of the chip's
        # VALU issue slots the step uses, from the committed counter pass (wave-level VALU instructions per step) and this run's step time.
        with open(issue_file) as f:
            vi = json.load(f).get(workload)
        # (the count belongs to the kernels it was taken from: where the file's steady-state reverb kernel is not the one this run
        # launched -- another build of the library, an experiment flag --, the step keeps its HBM roofline and says why.  ADVICE, round 3)
        counted = [k for k in (vi or {}).get("kernels", {}) if k.startswith("k_reverb_steady")]
        launched = batch.last_reverb_kernel
        if vi and counted and launched.startswith("k_reverb_steady") and launched not in counted:
            roofline["valu_issue_note"] = (f"profiles/valu_issue.json counts {counted[0]}, this run launched {launched}: no VALU-issue figure "
                                           "(refresh with scripts/pmc_configs.sh)")
            vi = None
        if vi:
            insts = vi["valu_wave_instructions_per_step"]
            peak = VALU_SIMDS * VALU_CLOCK_GHZ / VALU_CYCLES_PER_WAVE_INSTRUCTION   # G wave-instructions per second
            achieved = insts / (elapsed / args.steps) / 1e9
            roofline = {"bound": "valu-issue", "achieved": round(achieved, 1), "peak": round(peak, 1), "unit": "G wave-instructions/s",
                        "frac": round(achieved / peak, 4), "traffic": None,
                        "kernel": "all effect kernels of a step (VALU wave instructions of the step, profiles/valu_issue.json / step time)",
                        "kernel_us": roofline["kernel_us"], "launches_timed": args.steps, "valu_wave_instructions_per_step": insts,
                        "peak_note": f"{VALU_SIMDS} SIMDs x {VALU_CLOCK_GHZ} GHz / {VALU_CYCLES_PER_WAVE_INSTRUCTION} cycles per wave64 instruction",
                        "hbm": {k: roofline[k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch")}}
    pc, pk, pbest, pworst = batch.placement()
    roofline["delay_line_placement"] = {"chunks": pc, "candidates_probed": pk, "probe_us_kept": round(pbest, 2), "probe_us_slowest_seen": round(pworst, 2),
                                        "note": "the runtime keeps the fastest of a few candidate allocations for the delay lines (traffic-only probe; DESIGN 2)"}
    roofline["launch_plan_last_slot"] = {"ring_light": plan[0], "reverbs_proven_steady": plan[1], "reverbs_believed_steady": plan[2],
                                         "reverbs_general": plan[3]}

    total_frames = world * n * FRAMES * args.steps
    result = {
        "metric": METRIC if workload.startswith("config2") else METRICS[workload],
        "value": round(total_frames / elapsed / 1e6, 3),
        "unit": "Msamples/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": (WORKLOADS["config5"] if config5_main else WORKLOADS[workload]).format(n=n),
            "instances_per_gpu": n,
            "device_spin_up_ms": args.spin_up_ms,
            "frames_per_buffer": FRAMES,
            "parallelism": f"batch-split x{world}, no collectives",
            "devices": devices,
        },
        "roofline": roofline,
        "kernels": kernels,
    }
    # Consecutive calls of the timed region overlap on the device where the step is one steady-state launch (chained launches):
    # the launches of the roofline region carry events and run alone, one after the other -- `frac` is the kernel by itself; what the
    # chip moves with two launches in flight is the step's bytes over the step time
    step_gbs = bytes_per_step / (elapsed / args.steps) / 1e9
    result["roofline"]["chained_launches"] = {
        "calls_chained_in_timed_region": chained, "of": args.steps, "step_us": round(elapsed / args.steps * 1e6, 2),
        "achieved_over_step": round(step_gbs, 1), "frac_over_step": round(step_gbs / HBM_PEAK_GBS, 4),
        "note": "algorithmic bytes of a step / step time of the CLI-timed region (launch gaps included); --no-chain runs the region in plain stream order"}
    if workload == "config2" and not args.no_kernel_timing and not args.no_chain:
        # ... and in the steady state of a long run: CHAINED_RUN launches without a synchronisation, the time between the first's start and
        # the last's end on the host's clock over the launches (a launch ends every so often; each takes about twice that from start to end)
        for k in range(ROOFLINE_WARMUP):
            batch.mix_device(FRAMES, src[k % n_in].data_ptr(), dst.data_ptr())
        batch.synchronize()
        c0 = batch.chained_calls
        t0 = time.perf_counter()
        for k in range(CHAINED_RUN):
            batch.mix_device(FRAMES, src[k % n_in].data_ptr(), dst.data_ptr())
        batch.synchronize()
        interval = (time.perf_counter() - t0) / CHAINED_RUN
        run_gbs = bytes_per_step / interval / 1e9
        result["roofline"]["chained_launches"].update({
            "steady_state_interval_us": round(interval * 1e6, 2), "steady_state_launches": CHAINED_RUN,
            "steady_state_calls_chained": batch.chained_calls - c0, "achieved_steady_state": round(run_gbs, 1),
            "frac_steady_state": round(run_gbs / HBM_PEAK_GBS, 4)})
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file) and workload == "config2" and n == 4096 and not args.no_kernel_timing:
        with open(traffic_file) as f:
            t = json.load(f)
        if os.environ.get("OALSFX_TRAFFIC_REFRESH"):
            t = {}  # the counter passes that produce the next traffic.json run this very script (scripts/profile_pmc.sh)
        elif t.get("kernel") != roofline["kernel"]:
            # the committed counter run belongs to another kernel (an experiment flag, a preset that selects another build, a run whose
            # instances were not all proven yet): its bytes are not attached to this one
            result["roofline"]["traffic_note"] = (f"profiles/traffic.json was measured on {t.get('kernel')!r}, this run launched "
                                                  f"{roofline['kernel']!r}: no counter traffic for it (scripts/profile_pmc.sh refreshes the file)")
            t = {}
        result["roofline"]["traffic"] = t.get("hbm_bytes_per_launch")
        if t:
            result["roofline"]["traffic_source"] = t.get("source")

    if os.environ.get("OALSFX_DUMP_OUTPUT"):
        # rehearsal hook (tests/test_gpu_async_and_ranks.py): this rank's last output buffer, to compare with a single-process run
        import numpy as np
        np.save(os.environ["OALSFX_DUMP_OUTPUT"], dst.cpu().numpy())
    if args.host_io > 0 and world == 1:
        result["host_io"] = host_io_leg(batch, n, args.host_io)
    batch.close()
    del src, dst
    torch.cuda.empty_cache()

    if world == 1 and workload == "config2" and not config5_main and not args.preset_mix and args.instances == 0 and not args.no_other_configs and not args.in_process:
        # BASELINE.json's other single-GPU configurations in the same run, briefly (the headline stays configs[1]): configs[2], the 4-slot
        # chain, and configs[3], eleven effect types in one slot -- whole-step figures like `value`, inputs resident in HBM
        result["other_configs"] = {name: other_config_leg(name, 100, sharding, backend) for name in ("config3", "config4")}   # (100 steps whatever --steps says: a report beside the headline)

    if (args.config5 or distributed) and not config5_main and not args.no_config5:
        result["config5"] = config5_leg(rank, world, local_rank, min(args.steps, 100), sharding, backend)

    if distributed:
        dist.destroy_process_group()
    if rank == 0:
        # the CPU path in the same run (north_star), on rank 0's host cores, after the ranks have parted
        result["cpu_baseline"] = None if args.no_cpu_baseline else cpu_baseline()
        print(json.dumps(result), flush=True)


def host_io_leg(batch, n, steps):
    """PCIe-inclusive rates (never `value`; single-GPU runs only), pinned host buffers: every call synchronous
    (oalsfx_batch_mix), and pipelined across successive calls (oalsfx_batch_mix_async / oalsfx_batch_wait) where the library has it."""
    import ctypes as C
    import torch
    from oalsfxpp_amd import lib
    fp = C.POINTER(C.c_float)
    so = lib.load()
    depth = 3
    hsrc = [torch.empty(n, FRAMES, CHANNELS, dtype=torch.float32).uniform_(-1, 1).pin_memory() for _ in range(depth)]
    hdst = [torch.empty(n, FRAMES, CHANNELS, dtype=torch.float32).pin_memory() for _ in range(depth)]
    call = lambda k: so.oalsfx_batch_mix(batch._h, FRAMES, C.cast(hsrc[k % depth].data_ptr(), fp), C.cast(hdst[k % depth].data_ptr(), fp))
    for k in range(3):
        assert call(k)
    per_step = []
    t0 = time.perf_counter()
    for k in range(steps):
        t1 = time.perf_counter()
        assert call(k)
        per_step.append((time.perf_counter() - t1) * 1e3)
    dt = time.perf_counter() - t0
    per_step.sort()
    out = {"value": round(n * FRAMES * steps / dt / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
           "ms_per_step_min": round(per_step[0], 4), "ms_per_step_median": round(percentile(per_step, 0.5), 4), "ms_per_step_max": round(per_step[-1], 4),
           "note": "pinned host src/dst, H2D + kernels + D2H + sync per step, not overlapped"}
    if hasattr(so, "oalsfx_batch_mix_timed"):
        # where a step's time goes: the three legs as HIP events on the batch's stream see them, and the host's own clock around the call
        legs = (C.c_double * 3)()
        rows = []
        for k in range(steps):
            t1 = time.perf_counter()
            assert so.oalsfx_batch_mix_timed(batch._h, FRAMES, C.cast(hsrc[k % depth].data_ptr(), fp), C.cast(hdst[k % depth].data_ptr(), fp), legs)
            rows.append((legs[0], legs[1], legs[2], (time.perf_counter() - t1) * 1e6))
        med = lambda j: round(percentile(sorted(r[j] for r in rows), 0.5), 1)
        out["legs_us_median"] = {"h2d": med(0), "kernels": med(1), "d2h": med(2), "host_wall": med(3),
                                 "note": "oalsfx_batch_mix_timed; host_wall - (h2d + kernels + d2h) = queueing and the wake-up after hipStreamSynchronize"}
        out["legs_us_max"] = {"h2d": round(max(r[0] for r in rows), 1), "kernels": round(max(r[1] for r in rows), 1),
                              "d2h": round(max(r[2] for r in rows), 1), "host_wall": round(max(r[3] for r in rows), 1)}
    if hasattr(so, "oalsfx_batch_mix_async"):
        acall = lambda k: so.oalsfx_batch_mix_async(batch._h, FRAMES, C.cast(hsrc[k % depth].data_ptr(), fp), C.cast(hdst[k % depth].data_ptr(), fp))
        psteps = max(steps, 30)
        # (the library settles on three streams or one with the first forty-eight calls of a device's first batch that pipelines, DESIGN 4;
        # the timed calls come behind that)
        for k in range(52):
            assert acall(k)
        assert so.oalsfx_batch_wait(batch._h)
        t0 = time.perf_counter()
        for k in range(psteps):
            assert acall(k)     # waits by itself for the call that last used this staging slot
        assert so.oalsfx_batch_wait(batch._h)
        dt = time.perf_counter() - t0
        out["pipelined"] = {"value": round(n * FRAMES * psteps / dt / 1e6, 3), "unit": "Msamples/s", "ms_per_step": round(dt / psteps * 1e3, 4),
                            "steps": psteps, "note": "oalsfx_batch_mix_async x K then oalsfx_batch_wait: the copies of successive calls beside the kernels, "
                                                     "in the form the library's probe chose for this box"}
        if hasattr(so, "oalsfx_debug_host_pipeline"):
            form, us3, us2 = batch.host_pipeline()
            out["pipelined"]["form"] = {3: "three streams: H2D of call k+1 | kernels of call k | D2H of call k-1", 1: "one stream: H2D, kernels, D2H of a call in order, no wait in between"}.get(form, "probing")
            out["pipelined"]["probe_ms_per_step"] = {"three_streams": round(us3 * 1e-3, 4), "one_stream": round(us2 * 1e-3, 4)}
    return out


def other_config_leg(name, steps, sharding, backend):
    """`steps` steps of another BASELINE configuration (bench.py --workload <name> is the full report), timed like the main region."""
    from oalsfxpp_amd import desc, workloads
    from oalsfxpp_amd.api import Batch
    n = {"config3": 4096, "config4": 8192}[name]
    batch = Batch(n, desc.FMT_STEREO, 48000, workloads.effect_count(name), device_id=0)
    workloads.setup(batch, name)
    n_in = 4
    src, dst = resident_inputs(batch, n, 0, n_in)
    for k in range(8):   # (through the reverbs' start-up cross-fade, call by call: the device's report of what has settled is read back between calls)
        batch.mix_device(FRAMES, src[k % n_in].data_ptr(), dst.data_ptr())
        batch.synchronize()
    for k in range(40):
        batch.mix_device(FRAMES, src[k % n_in].data_ptr(), dst.data_ptr())
    before = batch.chained_calls
    elapsed = timed_region(batch, src, dst, n_in, 0, steps, sharding, backend)
    out = {"workload": WORKLOADS[name].format(n=n), "value": round(n * FRAMES * steps / elapsed / 1e6, 3), "unit": "Msamples/s", "steps": steps,
           "ms_per_step": round(elapsed / steps * 1e3, 5), "calls_chained": batch.chained_calls - before}
    batch.close()
    del src, dst
    import torch
    torch.cuda.empty_cache()
    return out


def config5_leg(rank, world, local_rank, steps, sharding, backend):
    """One GPU's share of BASELINE.json configs[4] (262144 EAX reverbs over 8 GPUs = 32768 per GPU, 28.75 GiB of delay lines each),
    timed like the main region."""
    from oalsfxpp_amd import desc, workloads
    from oalsfxpp_amd.api import Batch
    n = 32768
    batch = Batch(n, desc.FMT_STEREO, 48000, 1, device_id=local_rank)
    workloads.setup(batch, "config2", first_instance=rank * n)
    n_in = 4
    src, dst = resident_inputs(batch, n, rank, n_in)
    for k in range(16):
        batch.mix_device(FRAMES, src[k % n_in].data_ptr(), dst.data_ptr())
    batch.synchronize()
    for k in range(16):
        batch.mix_device(FRAMES, src[k % n_in].data_ptr(), dst.data_ptr())
    elapsed = timed_region(batch, src, dst, n_in, 32, steps, sharding, backend)
    # the kernel by itself: a region of its own behind the timed one (launches that carry events do not overlap with their neighbours,
    # so the timed region above, whose calls chain, carries none)
    batch.kernel_timing(TIMED_EVERY)
    for k in range(max(2 * TIMED_EVERY, min(steps, 40))):
        batch.mix_device(FRAMES, src[k % n_in].data_ptr(), dst.data_ptr())
    batch.synchronize()
    launches, ms = batch.kernel_timing_read(desc.EAX_REVERB)
    batch.kernel_timing(0)
    bracket_us = batch.event_overhead(100)
    kernel_us = ms / max(launches, 1) * 1e3 - bracket_us
    bytes_per_launch = BYTES_PER_FRAME * n * FRAMES
    achieved = bytes_per_launch / (kernel_us * 1e-6) / 1e9 if kernel_us > 0 else 0.0
    plan = batch.plan(0)
    out = {"workload": WORKLOADS["config5"].format(n=n), "instances_per_gpu": n, "instances_total": n * world,
           "value": round(world * n * FRAMES * steps / elapsed / 1e6, 3), "unit": "Msamples/s", "steps": steps, "warmup": 32,
           "ms_per_step": round(elapsed / steps * 1e3, 5),
           "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "kernel": batch.last_reverb_kernel, "kernel_us": round(kernel_us, 2), "launches_timed": launches,
                        "algorithmic_bytes_per_launch": bytes_per_launch, "reverbs_proven_steady": plan[1]}}
    batch.close()
    return out


if __name__ == "__main__":
    main()
