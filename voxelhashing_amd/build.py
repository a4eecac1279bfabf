"""Builds libvoxelhashing_amd.so (HIP kernels + C ABI + host classes) in-tree
for gfx950 with hipcc.  No GPU is needed to build.

Flags that are part of the numerical contract (DESIGN.md "Numerics"):
  -ffp-contract=off   no FMA contraction: block ids / pixel ids are float->int cliffs
  -fhip-fp32-correctly-rounded-divide-sqrt   IEEE fp32 division and sqrt
  (no -ffast-math)
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libvoxelhashing_amd.so")
SOURCES = ["vh_kernels.hip", "vh_probe.hip", "vh_host.cpp", "vh_chunk_grid.cpp", "vh_marching_cubes.cpp", "vh_sensor.cpp", "vh_sensor_data.cpp", "vh_params.cpp", "vh_tracking.cpp", "vh_reconstruction.cpp", "vh_c_api.cpp"]
HEADERS = ["vh_device.hpp", "vh_host_util.hpp", "vh_stage_timer.hpp", "vh_handles.hpp",
           os.path.join(ROOT, "include", "vh_types.h"), os.path.join(ROOT, "include", "vh_api.h"),
           os.path.join(ROOT, "include", "vh.hpp"), os.path.join(ROOT, "include", "vh_mc_tables.h")]
ARCH = "gfx950"


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def flags():
    return ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
            "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wall", "-Wno-unused-function",
            "-I", os.path.join(ROOT, "include")]


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    deps.append(os.path.abspath(__file__))
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False, extra=(), out=None):
    """out: build a variant (extra flags, e.g. -DVH_KNOCKOUT=1 for a measurement build) into another file; its objects
    go to a directory of their own and the shipped library is left alone"""
    if out is not None:
        return build_variant(out, list(extra), verbose)
    if not force and up_to_date():
        return LIB
    cc = hipcc()
    lib_t = os.path.getmtime(LIB) if os.path.exists(LIB) else 0.0
    hdr_t = max(os.path.getmtime(h if os.path.isabs(h) else os.path.join(CSRC, h)) for h in HEADERS)
    hdr_t = max(hdr_t, os.path.getmtime(os.path.abspath(__file__)))
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        # an object is reused when it is newer than its source and every header (and no extra flags are given)
        if not force and not extra and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), hdr_t):
            continue
        jobs.append([cc] + flags() + list(extra) + ["-c", src, "-o", obj])
    procs = []
    for cmd in jobs:  # the translation units compile side by side (the kernels' one takes the longest)
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    failed = [cmd for cmd, p in procs if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-lpthread", "-lz"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def build_variant(out, extra, verbose=False):
    cc = hipcc()
    objdir = out + ".objs"
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for s in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(s)[0] + ".o")
        objs.append(obj)
        cmd = [cc] + flags() + extra + ["-c", os.path.join(CSRC, s), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    failed = [cmd for cmd, p in procs if p.wait() != 0]
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    subprocess.check_call([cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs + ["-lpthread", "-lz"])
    return out


if __name__ == "__main__":
    if "--out" in sys.argv:  # python -m voxelhashing_amd.build --out scratch/lib_ko1.so -DVH_KNOCKOUT=1
        print(build(out=os.path.abspath(sys.argv[sys.argv.index("--out") + 1]), extra=[a for a in sys.argv[1:] if a.startswith("-D")], verbose=True))
    else:
        build(force="--force" in sys.argv, verbose=True)
        print(LIB)
