#!/usr/bin/env python3
"""Static instruction mix of one kernel of a .hip source (no GPU needed): compiles the device side to assembly with the build's flags and
counts the instructions of the kernel whose demangled name contains the given text.
    python3 scripts/isa_mix.py "k_reverb_steady_coop<2, 4, false, false, false, false, false, true, false, false, false, 0>" [source] [--save out.s]"""
import collections, os, re, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1]
src = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "oalsfxpp_amd/csrc/hip/reverb.hip"
save = sys.argv[sys.argv.index("--save") + 1] if "--save" in sys.argv else None
asm = os.environ.get("ISA_ASM")
if not asm:
    asm = os.path.join(tempfile.mkdtemp(), "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", f"-I{R}/include", f"-I{R}/oalsfxpp_amd/csrc/host",
                    f"-I{R}/oalsfxpp_amd/csrc/hip", "--offload-arch=gfx950", "-x", "hip", "--cuda-device-only", "-S", os.path.join(R, src), "-o", asm],
                   check=True, stderr=subprocess.DEVNULL)
s = open(asm).read()
parts = re.split(r"\n(_Z\w+):[^\n]*\n", s)
for name, body in zip(parts[1::2], parts[2::2]):
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("oalsfx_hip::", "")
    if want not in dn:
        continue
    body = body.split(".Lfunc_end")[0]
    ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
    c = collections.Counter(ins)
    groups = collections.Counter()
    for k, v in c.items():
        groups["valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_") else
               "vmem" if k.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"] += v
    print(dn.split("(")[0], len(ins), dict(groups))
    print("  lane spills (v_readlane / v_writelane):", c["v_readlane_b32"] + c["v_writelane_b32"], " v_mov_b32:", c["v_mov_b32_e32"] + c["v_mov_b32_e64"],
          " packed fp32:", sum(v for k, v in c.items() if k.startswith("v_pk_")))
    for k, v in c.most_common(40):
        print(f"  {v:5d} {k}")
    if save:
        open(save, "w").write(body)
