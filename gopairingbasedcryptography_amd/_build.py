"""Compile libgpbc_bn254.so for gfx950 with hipcc (in-tree, next to this file)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgpbc_bn254.so")
SOURCES = ["gpbc_bn254.hip"]
HEADERS = ["fe29.cuh", "tower29.cuh", "curve29.cuh", "pairing29.cuh", "bn254_constants.cuh", "bn254_constants29.cuh"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps.append(os.path.join(HERE, "..", "include", "gpbc_bn254.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> gopairingbasedcryptography_amd/libgpbc_bn254.so; returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
