"""Compile libgpbc_bn254.so for gfx950 with hipcc (in-tree, next to this file).

Five translation units (csrc/gpbc_core.hip, gpbc_pairing.hip, gpbc_curve.hip, gpbc_wire.hip, gpbc_msm.hip) are compiled in parallel and
linked into one shared library; every unit carries its own device code (no relocatable device code is needed: kernels are
launched from the unit that defines them)."""
import hashlib
import os
import shutil
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgpbc_bn254.so")
STAMP = os.path.join(HERE, "libgpbc_bn254.buildhash")
SOURCES = ["gpbc_core.hip", "gpbc_pairing.hip", "gpbc_curve.hip", "gpbc_wire.hip", "gpbc_msm.hip"]
HEADERS = ["gpbc_common.hpp", "fe29.hip.hpp", "tower29.hip.hpp", "tower29_pair.hip.hpp", "curve29.hip.hpp", "pairing29.hip.hpp", "pairing29_pair.hip.hpp", "wide29.hip.hpp", "curve29_quad.hip.hpp", "curve29_oct.hip.hpp",
           "wire29.hip.hpp", "h2c29.hip.hpp", "xmd29.hip.hpp", "msm29.hip.hpp", "bn254_constants.hip.hpp", "bn254_constants29.hip.hpp"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"]
# host-only measurement program over the C ABI (bench.py runs it: calls/s of concurrent one-element calls); built next to the library
CALLS_SRC = os.path.join(HERE, "..", "tools", "concurrent_calls.cpp")
CALLS_EXE = os.path.join(HERE, "gpbc_concurrent_calls")


def _source_hash():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(HERE, "..", "include", "gpbc_bn254.h"), CALLS_SRC]
    for d in deps:
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale():
    """Content hash, not mtimes: the GPU box receives a copy of the tree whose timestamps are not the build's."""
    if not os.path.exists(LIB) or not os.path.exists(STAMP) or not os.path.exists(CALLS_EXE):
        return True
    with open(STAMP) as f:
        return f.read().strip() != _source_hash()


def _check_isa(tmp):
    """Refuse device code that contains v_subrev_*_dpp.  The compiler folds `a - swap(b)` (swap = a single-use DPP lane move) into
    v_subrev_u32_dpp assuming src1 - dpp(src0); gfx950 computes dpp(src1) - src0 (tools/subdpp_probe.hip: every lane wrong, for the
    compiler's output and for the hand-written instruction).  csrc/tower29_pair.hip.hpp subtracts on the sending lane instead; this
    check keeps a later edit from re-introducing the pattern silently (the host interval harness has no DPP and cannot see it)."""
    import glob
    import re
    bad = []
    for s in SOURCES:
        unit = s.replace(".hip", "")
        files = glob.glob(os.path.join(tmp, unit, "*gfx950*.s"))
        if not files:                                   # a check that scanned nothing has checked nothing
            raise RuntimeError("no gfx950 ISA file (-save-temps) found for %s under %s: the v_subrev_*_dpp fence cannot run" % (s, os.path.join(tmp, unit)))
        for f in files:
            with open(f, errors="replace") as fh:
                n = len(re.findall(r"^\s+v_subb?rev\w*_dpp\b", fh.read(), flags=re.M))
            if n:
                bad.append("%s: %d" % (os.path.basename(f), n))
    if bad:
        raise RuntimeError("device ISA contains v_subrev_*_dpp (wrong on gfx950, see _build._check_isa): " + ", ".join(bad))


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> gopairingbasedcryptography_amd/libgpbc_bn254.so; returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory(prefix="gpbc_build_") as tmp:
        procs = []
        for s in SOURCES:
            unit = os.path.join(tmp, s.replace(".hip", ""))
            os.makedirs(unit)
            obj = os.path.join(unit, s.replace(".hip", ".o"))
            cmd = [hipcc] + FLAGS + ["-save-temps", "-c", os.path.join(CSRC, s), "-o", obj]      # temps (the device ISA among them) land in cwd = unit
            if verbose:
                print(" ".join(cmd))
            log = open(os.path.join(unit, "stderr.txt"), "w+")
            procs.append((cmd, obj, subprocess.Popen(cmd, cwd=unit, stderr=None if verbose else log), log))
        objs = []
        for cmd, obj, p, log in procs:
            if p.wait() != 0:
                for _, _, q, _ in procs:
                    if q.poll() is None:
                        q.kill()
                log.seek(0)
                raise RuntimeError("%s failed (exit %d):\n%s" % (" ".join(cmd), p.returncode, log.read()[-8000:]))
            objs.append(obj)
        _check_isa(tmp)
        link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(link))
        subprocess.check_call(link)
    calls = ["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(HERE, "..", "include"), CALLS_SRC, "-L" + HERE, "-lgpbc_bn254",
             "-Wl,-rpath,$ORIGIN", "-o", CALLS_EXE]
    if verbose:
        print(" ".join(calls))
    subprocess.check_call(calls)
    with open(STAMP, "w") as f:
        f.write(_source_hash() + "\n")
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
