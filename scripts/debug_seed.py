"""Replays one seed of tests/test_gpu_parity.py::test_random_scripts and prints, per mix call, the launch plan of every slot and
the steady-state kernel that ran: python scripts/debug_seed.py <seed>"""
import random, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
from oalsfxpp_amd import desc
from oalsfxpp_amd.api import Batch, BatchError
from oalsfxpp_amd.workloads import random_effect
from oracle import oracle as orc

seed = int(sys.argv[1])
rng = random.Random(1234 + seed)
fmt = rng.choice([desc.FMT_MONO, desc.FMT_STEREO, desc.FMT_STEREO, desc.FMT_QUAD, desc.FMT_5POINT1, desc.FMT_7POINT1])
rate = rng.choice([22050, 44100, 48000, 48000, 96000])
slots = rng.randint(1, 4)
n = 6
types = list(range(12))
setups = [[(s, random_effect(rng, rng.choice(types))) for s in range(slots)] for _ in range(n)]
script = []
for _ in range(14):
    r = rng.random()
    if r < 0.55:
        script.append(("mix", rng.choice([1, 2, 63, 64, 64, 128, 256, 256, 256, 300, 2048 + 17])))
    elif r < 0.75:
        script.append(("set", rng.randrange(n), rng.randrange(slots), random_effect(rng, rng.choice(types))))
        script.append(("apply",))
    elif r < 0.9:
        script.append(("send", rng.randrange(n), rng.randint(-1, slots - 1), rng.uniform(0.2, 1.0), rng.choice([1.0, rng.uniform(0.1, 1.0)]),
                       rng.choice([1.0, rng.uniform(0.1, 1.0)])))
        script.append(("apply",))
    else:
        script.append(("apply",))
script += [("mix", 256), ("mix", 256)]
print("format", fmt, "rate", rate, "slots", slots)
with Batch(n, fmt, rate, slots) as b:
    for i, eff in enumerate(setups):
        print(i, [(s, e.type) for s, e in eff])
        for slot, e in eff:
            b.set_effect(slot, e, first=i, count=1)
    b.apply_changes()
    k = 0
    for op in script:
        if op[0] == "set":
            print("set", op[1], op[2], op[3].type)
            b.set_effect(op[2], op[3], first=op[1], count=1)
        elif op[0] == "send":
            print("send", op[1:])
            b.set_send_props(op[2], op[3], op[4], op[5], first=op[1], count=1)
        elif op[0] == "apply":
            b.apply_changes()
        else:
            frames = op[1]
            print("mix", frames, "plans", [b.plan(s) for s in range(slots)], flush=True)
            for i in range(n):
                for s_ in range(slots):
                    p, st = b.read_slot(i, s_)
                    if p.type in (desc.REVERB, desc.EAX_REVERB):
                        r, q = p.u.reverb, st.u.reverb
                        ch = b.channels
                        off = [(a, c) for a in range(4) for c in range(ch) if r.early_pan[a][c] != q.early_cur_gain[a][c] or r.late_pan[a][c] != q.late_cur_gain[a][c]]
                        print(f"   before: instance {i} slot {s_}: seen {st.seen_seq} of {p.update_seq}, fade {q.fade_count}, mod_filter {q.mod_filter}, gains off target: {len(off)}"
                              + (f" e.g. {r.early_pan[off[0][0]][off[0][1]]!r} vs {q.early_cur_gain[off[0][0]][off[0][1]]!r} / {r.late_pan[off[0][0]][off[0][1]]!r} vs {q.late_cur_gain[off[0][0]][off[0][1]]!r}" if off else ""))
            x = np.stack([orc.synth(1000 + i, k, frames * b.channels).reshape(frames, b.channels) for i in range(n)])
            try:
                b.mix(x)
            except BatchError as e:
                print("   FAILED:", e, "| kernel:", b.last_reverb_kernel)
                for i in range(n):
                    for s in range(slots):
                        p, st = b.read_slot(i, s)
                        if p.type in (desc.REVERB, desc.EAX_REVERB):
                            r = p.u.reverb
                            print(f"   instance {i} slot {s}: early_tap {list(r.early_tap)} ap {list(r.early_ap_off)} line {list(r.early_line_off)} late_tap {list(r.late_tap)} feed {r.late_feed_tap} "
                                  f"ap {list(r.late_ap_off)} line {list(r.late_line_off)} mod_depth {r.mod_depth}")
                break
            print("   kernel:", b.last_reverb_kernel)
            k += 1
