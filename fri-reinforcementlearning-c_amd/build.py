"""Builds the in-tree native libraries of the package (hipcc cross-compiles gfx950 without a GPU).

    python fri-reinforcementlearning-c_amd/build.py        # or frirl_amd.build()

Outputs (git-ignored, shipped to the GPU box by gpurun):
    lib/libfrirl_hip.so   hand-written HIP kernels + the C ABI declared in include/frirl_hip.h
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
HIP_LIB = os.path.join(LIBDIR, "libfrirl_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
             "-ffp-contract=off",      # no IMPLICIT contraction: distances, snaps and env dynamics keep separate mul / add like the
             # reference's vmulpd/vaddpd (bit-exact); the Q sweeps use EXPLICIT __fma_rn inside their <= 1e-6 contract
             "-Wall", "-Wno-unused-function"]


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_hip(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    if not force and not _stale(HIP_LIB, srcs + hdrs):
        return HIP_LIB
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = [f for f in HIP_FLAGS if f != "-shared"] + ["-I", os.path.join(ROOT, "include")]
    jobs = []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + flags + ["-c", src, "-o", obj]
            if verbose:
                print("+", " ".join(cmd))
            jobs.append((src, subprocess.Popen(cmd)))
    failed = [src for src, pr in jobs if pr.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for " + ", ".join(failed))
    objs = [os.path.join(objdir, os.path.basename(src)[:-4] + ".o") for src in srcs]
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs + ["-ldl", "-lpthread"]
    if verbose:
        print("+", " ".join(cmd))
    subprocess.run(cmd, check=True)
    return HIP_LIB


HOST_DIR = os.path.join(PKG, "host")
DROPIN_LIB = os.path.join(LIBDIR, "libfrirl_dropin.so")
DEMO_BIN = os.path.join(LIBDIR, "frirl_demo")
CC = os.environ.get("CC", "gcc")
# ANSI C host: no FMA contraction, same optimisation level as the reference (CMakeLists.txt:6)
HOST_CFLAGS = ["-O2", "-std=gnu99", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-parameter"]


def build_host(force=False, verbose=False):
    """libfrirl_dropin.so: the reference's five_* / FIVE_* / frirl_* C API on top of libfrirl_hip.so; + the demo driver."""
    srcs = [os.path.join(HOST_DIR, f) for f in ("five_host.c", "frirl_host.c", "frirl_io.c", "demo_envs.c")]
    deps = srcs + glob.glob(os.path.join(HOST_DIR, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")) + [HIP_LIB]
    inc = ["-I", os.path.join(ROOT, "include"), "-I", HOST_DIR]
    if force or _stale(DROPIN_LIB, deps):
        cmd = [CC] + HOST_CFLAGS + inc + ["-shared", "-o", DROPIN_LIB] + srcs + ["-L", LIBDIR, "-lfrirl_hip", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"]
        if verbose:
            print("+", " ".join(cmd))
        subprocess.run(cmd, check=True)
    demo_src = os.path.join(HOST_DIR, "demo.c")
    if force or _stale(DEMO_BIN, [demo_src, DROPIN_LIB]):
        cmd = [CC] + HOST_CFLAGS + inc + ["-o", DEMO_BIN, demo_src, "-L", LIBDIR, "-lfrirl_dropin", "-lfrirl_hip", "-Wl,-rpath,$ORIGIN", "-lm"]
        if verbose:
            print("+", " ".join(cmd))
        subprocess.run(cmd, check=True)
    return [DROPIN_LIB, DEMO_BIN]


def build_all(force=False, verbose=False):
    return [build_hip(force, verbose)] + build_host(force, verbose)


if __name__ == "__main__":
    for p in build_all(force="--force" in sys.argv, verbose=True):
        print("built", p)
