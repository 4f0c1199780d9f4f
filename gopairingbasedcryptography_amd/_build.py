"""Compile libgpbc_bn254.so for gfx950 with hipcc (in-tree, next to this file)."""
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgpbc_bn254.so")
STAMP = os.path.join(HERE, "libgpbc_bn254.buildhash")
SOURCES = ["gpbc_bn254.hip"]
HEADERS = ["fe29.cuh", "tower29.cuh", "tower29_pair.cuh", "curve29.cuh", "pairing29.cuh", "pairing29_pair.cuh", "wire29.cuh", "h2c29.cuh", "bn254_constants.cuh", "bn254_constants29.cuh"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC"]


def _source_hash():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(HERE, "..", "include", "gpbc_bn254.h")]
    for d in deps:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale():
    """Content hash, not mtimes: the GPU box receives a copy of the tree whose timestamps are not the build's."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _source_hash()


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> gopairingbasedcryptography_amd/libgpbc_bn254.so; returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(STAMP, "w") as f:
        f.write(_source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
