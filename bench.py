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
  cpu_baseline  the CPU oracle (oracle/liboracle.so, kind "port") timed on this host's cores on a bounded
                sample of the same workload (rank 0, N=1 only)
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
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
METRIC = "Msamples/sec EAX reverb, 256-frame buffers, batch=4096; % HBM roofline"


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
    return {
        "value": round(instances * buffers * FRAMES / t / 1e6, 3),
        "unit": "Msamples/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{instances} EAX-reverb instances x {buffers} buffers of {FRAMES} stereo frames after {warm} warm-up buffers, "
                  f"{threads} threads, oracle/liboracle.so (-O2 -ffp-contract=off), {t:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--instances", type=int, default=4096, help="instances per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--preset-mix", action="store_true", help="robustness run: instance i uses EFX preset i %% 113")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from oalsfxpp_amd import desc, lib, sharding
    from oalsfxpp_amd.api import Batch

    rank, world, local_rank = sharding.env_rank_world()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    distributed = world > 1
    torch.cuda.set_device(local_rank)
    if distributed:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    n = args.instances
    batch = Batch(n, desc.FMT_STEREO, 48000, 1, device_id=local_rank)
    if args.preset_mix:
        effects = []
        for i in range(n):
            e = lib.effect_defaults(desc.EAX_REVERB)
            e.props.reverb = lib.preset(i % lib.preset_count())[1]
            effects.append(e)
        batch.set_effect(0, effects)
    else:
        batch.set_effect_type(0, desc.EAX_REVERB)
    batch.apply_changes()

    # inputs resident in HBM: a ring of pre-generated buffers, one output buffer
    n_in = 8
    floats = n * FRAMES * CHANNELS
    src = [torch.empty(floats, dtype=torch.float32, device="cuda") for _ in range(n_in)]
    dst = torch.empty(floats, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for k, s in enumerate(src):
        batch.fill_synthetic(FRAMES, k + 1000 * rank, s.data_ptr())
    batch.synchronize()

    def step(k):
        batch.mix_device(FRAMES, src[k % n_in].data_ptr(), dst.data_ptr())

    for k in range(args.warmup):
        step(k)
    batch.synchronize()
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()

    batch.kernel_timing(True)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    batch.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sharding.barrier()
    torch.cuda.synchronize()
    elapsed = sharding.max_over_ranks(elapsed, device="cuda")

    launches, kernel_ms = batch.kernel_timing_read(desc.EAX_REVERB)
    batch.kernel_timing(False)
    avg_kernel_s = kernel_ms / max(launches, 1) / 1e3
    frames_per_launch = n * FRAMES
    achieved_gbs = BYTES_PER_FRAME * frames_per_launch / avg_kernel_s / 1e9 if avg_kernel_s > 0 else 0.0

    total_frames = world * n * FRAMES * args.steps
    result = {
        "metric": METRIC,
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
            "workload": f"{n} independent EAX-reverb instances per GPU, stereo, 48 kHz, 256-frame buffers, 1 slot, "
                        + ("EFX preset i%113 per instance" if args.preset_mix else "default properties")
                        + " (BASELINE.json configs[1])",
            "instances_per_gpu": n,
            "frames_per_buffer": FRAMES,
            "parallelism": f"batch-split x{world}, no collectives",
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved_gbs, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved_gbs / HBM_PEAK_GBS, 4),
            "traffic": None,
            "kernel": "k_reverb_steady_coop<2,4>" if not args.preset_mix else "k_reverb_steady_coop<2,4> + k_reverb<2>",
            "kernel_us": round(avg_kernel_s * 1e6, 2),
            "launches_timed": launches,
            "algorithmic_bytes_per_launch": BYTES_PER_FRAME * frames_per_launch,
        },
    }
    traffic_file = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(traffic_file) and not args.preset_mix and n == 4096:
        with open(traffic_file) as f:
            t = json.load(f)
        result["roofline"]["traffic"] = t.get("hbm_bytes_per_launch")
        result["roofline"]["traffic_source"] = t.get("source")

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline()
    elif rank == 0:
        result["cpu_baseline"] = None

    batch.close()
    if distributed:
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
