#!/usr/bin/env python3
"""Build libidee_hip.so for gfx950 in-tree (idee_amd/libidee_hip.so).  hipcc cross-compiles
without a GPU.  Usage: python idee_amd/csrc/build.py [--force] [--save-temps]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
OUT = os.environ.get("IDEE_BUILD_OUT") or os.path.join(PKG, "libidee_hip.so")
SRCS = ["api.cpp", "swin_mfma.hip", "conv_mfma.hip", "conv_bf16.hip", "pointwise.hip"]
DEPS = ["common.h", "conv_common.h", os.path.join("..", "..", "include", "idee_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-ffp-contract=off"]
# A/B builds of kernel variants (tuning macros such as -DIDEE_RD_DEPTH=2): IDEE_EXTRA_FLAGS adds compiler flags and moves the
# objects to their own directory, IDEE_BUILD_OUT names the library; bench / tests pick it up through IDEE_HIP_LIB
EXTRA = os.environ.get("IDEE_EXTRA_FLAGS", "").split()
OBJDIR = "_obj" if not EXTRA else "_obj_" + "".join(c if c.isalnum() else "_" for c in " ".join(EXTRA))


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(os.path.join(HERE, d)) > t for d in [src] + DEPS + ["build.py"])


def build(force=False, save_temps=False, verbose=True):
    os.makedirs(os.path.join(HERE, OBJDIR), exist_ok=True)
    jobs = []
    for s in SRCS:
        obj = os.path.join(HERE, OBJDIR, s.rsplit(".", 1)[0] + ".o")
        if force or _stale(obj, s):
            cmd = [HIPCC, "-x", "hip", "-c", os.path.join(HERE, s), "-o", obj] + FLAGS + EXTRA
            if save_temps:
                cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
            jobs.append(cmd)
    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.join(HERE, OBJDIR))
        return r
    with ThreadPoolExecutor(max_workers=4) as ex:
        for r in ex.map(run, jobs):
            if r.returncode != 0 or (save_temps and r.stderr):
                sys.stderr.write(r.stdout + r.stderr)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed")
    objs = [os.path.join(HERE, OBJDIR, s.rsplit(".", 1)[0] + ".o") for s in SRCS]
    if jobs or not os.path.exists(OUT):
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", OUT] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
    print(OUT)
