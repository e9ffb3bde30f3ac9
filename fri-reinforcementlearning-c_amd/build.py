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
             "-ffp-contract=off",      # reference uses separate vmulpd/vaddpd: no FMA contraction anywhere
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
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", HIP_LIB] + objs
    if verbose:
        print("+", " ".join(cmd))
    subprocess.run(cmd, check=True)
    return HIP_LIB


def build_all(force=False, verbose=False):
    return [build_hip(force, verbose)]


if __name__ == "__main__":
    for p in build_all(force="--force" in sys.argv, verbose=True):
        print("built", p)
