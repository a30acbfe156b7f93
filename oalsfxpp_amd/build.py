"""In-tree build of liboalsfx_hip.so (host update path + batch runtime + gfx950 kernels).

    python -m oalsfxpp_amd.build        # or: __graft_entry__.build()

hipcc cross-compiles for gfx950 without a GPU present.  Flags that matter for parity:
-ffp-contract=off on host and device (the reference's arithmetic is unfused), no fast-math.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
# OALSFX_BUILD_TAG=<tag> (with OALSFX_EXTRA_FLAGS=-D...) builds an experiment variant beside the product for same-box A/B runs
# (scripts/ab_libs.py): objects in build/obj_<tag>, library ab/liboalsfx_hip_<tag>.so
_TAG = os.environ.get("OALSFX_BUILD_TAG", "")
OUT = os.path.join(ROOT, "ab", f"liboalsfx_hip_{_TAG}.so") if _TAG else os.path.join(CSRC, "liboalsfx_hip.so")
OBJ_DIR = os.path.join(ROOT, "build", "obj_" + _TAG if _TAG else "obj")

HOST_SOURCES = ["host/props.cpp", "host/panning.cpp", "host/update.cpp", "host/hostabi.cpp", "host/api.cpp", "host/api_array.cpp", "host/group.cpp"]
HIP_SOURCES = ["hip/batch.cpp", "hip/reverb.hip", "hip/support_kernels.hip", "hip/wave_effects.hip"]

COMMON = ["-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-Wall", "-Wextra", "-Wno-unused-parameter",
          "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(CSRC, "host"), "-I" + os.path.join(CSRC, "hip")]
DEVICE = ["--offload-arch=gfx950", "-x", "hip"]  # fp32 denormals stay on (the gfx9 default): the reference computes with them


# per-source additions
EXTRA = {
    "hip/wave_effects.hip": ["-fno-slp-vectorize"],
    "hip/reverb.hip": ["-fno-slp-vectorize"],
}


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in (src,) + tuple(extra))


def build_all(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = _hipcc()
    headers = []
    for d in (os.path.join(ROOT, "include"), os.path.join(CSRC, "host"), os.path.join(CSRC, "hip")):
        headers += [os.path.join(d, f) for f in os.listdir(d) if f.endswith((".h", ".hpp", ".inc"))]
    objs = []
    procs = []
    for rel in HOST_SOURCES + HIP_SOURCES:
        src = os.path.join(CSRC, rel)
        obj = os.path.join(OBJ_DIR, rel.replace("/", "_") + ".o")
        objs.append(obj)
        if not (force or _newer(src, obj, headers)):
            continue
        cmd = [hipcc] + COMMON + (DEVICE if rel in HIP_SOURCES else []) + EXTRA.get(rel, []) + os.environ.get("OALSFX_EXTRA_FLAGS", "").split() + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((rel, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for rel, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"---- {rel} ----\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed")
    if force or procs or not os.path.exists(OUT):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return OUT


TOOL_SRC = os.path.join(ROOT, "tools", "oalsfx_wav.cpp")
TOOL_OUT = os.path.join(ROOT, "tools", "oalsfx_wav")


def build_tools(force=False):
    """tools/oalsfx_wav: the WAV command-line program over oalsfxpp::Api, linked to the in-tree library."""
    if not (force or _newer(TOOL_SRC, TOOL_OUT, (OUT, os.path.join(ROOT, "include", "oalsfxpp.h")))):
        return TOOL_OUT
    libdir = os.path.dirname(OUT)
    subprocess.run(["g++", "-std=c++14", "-O2", "-I", os.path.join(ROOT, "include"), TOOL_SRC, "-L", libdir, "-loalsfx_hip",
                    "-Wl,-rpath,$ORIGIN/../oalsfxpp_amd/csrc", "-o", TOOL_OUT], check=True)
    return TOOL_OUT


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
    if not _TAG:
        print(build_tools(force="--force" in sys.argv))
