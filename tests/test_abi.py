"""CPU tests of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every symbol
include/gpbc_bn254.h declares, and fails loudly (no CPU fallback) when no GPU is present."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib():
    from gopairingbasedcryptography_amd import _build, _lib
    _build.build_library()
    return _lib.load()


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gpbc_bn254.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gpbc_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(lib):
    syms = declared_symbols()
    assert len(syms) >= 30
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_python_binding_covers_header(lib):
    from gopairingbasedcryptography_amd import _lib
    assert sorted(_lib.EXPORTS) == declared_symbols()


def test_generators_match_oracle():
    import bn254_py as o
    from gopairingbasedcryptography_amd import bn254
    g1, g2 = bn254.generators()
    assert g1.tobytes() == o.g1_to_bytes(o.G1_GEN)
    assert g2.tobytes() == o.g2_to_bytes(o.G2_GEN)
    assert bn254.R_ORDER == o.R


def test_no_cpu_fallback(lib):
    """Without a GPU every compute entry must fail with an error, never compute on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gopairingbasedcryptography_amd import bn254, EngineError
    g1, g2 = bn254.generators()
    with pytest.raises(EngineError):
        bn254.pair_batch(g1, g2)
    with pytest.raises(EngineError):
        bn254.g1_scalar_mul(g1, [5])
    out = np.zeros(384, dtype=np.uint8)
    rc = lib.gpbc_pair_batch(g1.ctypes.data, g2.ctypes.data, 1, out.ctypes.data)
    assert rc < 0 and not out.any()
    assert b"" != lib.gpbc_last_error()


def test_product_never_imports_oracle():
    """The product package must not reference oracle/ (parity claims depend on it)."""
    pkg = os.path.join(ROOT, "gopairingbasedcryptography_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hip.hpp", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_lib" not in text and "bn254_py" not in text and "bn254_oracle" not in text, f


def test_cpp_host_mirror_compiles_and_fails_loudly_without_gpu(lib, tmp_path):
    """The C++ mirror of the gnark surface links against the C ABI; without a GPU it must raise, not compute."""
    import subprocess
    import torch
    exe = str(tmp_path / "test_bls_flow")
    pkg = os.path.join(ROOT, "gopairingbasedcryptography_amd")
    subprocess.check_call(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_bls_flow.cpp"),
                           "-L" + pkg, "-lgpbc_bn254", "-Wl,-rpath," + pkg, "-o", exe])
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked test")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "no CPU fallback" in out.stderr


def test_integration_doc_uses_only_declared_symbols():
    """Every gpbc_* name in INTEGRATION.md (the cgo shim a maintainer would add, and the entry map) is declared in the
    header, and every declared entry point appears in the document's map."""
    import re
    header = open(os.path.join(ROOT, "include", "gpbc_bn254.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    declared = set(re.findall(r"\b(gpbc_[a-z0-9_]+)\s*\(", header))
    used = set(re.findall(r"\b(gpbc_[a-z0-9_]+)\b", doc)) - {"gpbc_bn254", "gpbc_status", "gpbc_fixed_base"}
    expand = set()
    for name in used:                                  # the map writes families as `gpbc_x(_dev)` or "gpbc_x_*"
        expand.add(name)
    unknown = {n for n in expand if n not in declared and not any(d.startswith(n) for d in declared)}
    assert not unknown, unknown
    missing = {d for d in declared if d not in doc and not any(d.startswith(u) and u != d for u in used)}
    assert not missing, missing


def test_wrapper_rejects_malformed_device_arguments():
    """The torch (HBM-resident) paths hand raw data_ptr()s to the C ABI, so sizes, dtypes and devices are checked on the host
    first (a wrong size would be an out-of-bounds device access).  Runs without a GPU: the checks come before any C call."""
    import torch
    from gopairingbasedcryptography_amd import bn254
    z = lambda n: torch.zeros(n, dtype=torch.uint8)
    P, Q = z(4 * 64), z(4 * 128)
    bad = [
        lambda: bn254.pair_batch(P, z(3 * 128)),                              # Q shorter than P
        lambda: bn254.pair_batch(P, Q, out=z(3 * 384)),                       # caller's out too small
        lambda: bn254.pair_batch(P, Q.to(torch.int8)),                        # dtype
        lambda: bn254.pair_batch(P, Q),                                       # right sizes, but host tensors: not CUDA
        lambda: bn254.miller_loop(P, z(5 * 128)),
        lambda: bn254.multi_pair(P, Q, [0, 4], out=z(2 * 384)),
        lambda: bn254.multi_pair(P, Q, torch.tensor([0, 4], dtype=torch.int32)),   # device table read as uint64
        lambda: bn254.g1_scalar_mul(z(64), z(4 * 32), out=z(3 * 64)),
        lambda: bn254.g1_scalar_mul(z(2 * 64), z(4 * 32)),                    # 2 bases for 4 scalars
        lambda: bn254.g2_scalar_mul(np.zeros(128, dtype=np.uint8), z(32)),    # host bases with device scalars
        lambda: bn254.gt_exp(z(4 * 384), z(3 * 32)),                          # one exponent short
        lambda: bn254.gt_exp(z(4 * 384), z(4 * 32), out=z(384)),
        lambda: bn254.gt_mul(z(4 * 384), z(3 * 384)),
        lambda: bn254.gt_div(z(4 * 384), z(5 * 384)),
        lambda: bn254.g1_scalar_mul_sum(z(3 * 64), z(4 * 32)),
        lambda: bn254.multi_pair_fixed_q(z(6 * 64), z(4 * 128)),              # 6 points are not a multiple of the 4-point list
        lambda: bn254.g1_unmarshal(z(65), elem_bytes=32),
        lambda: bn254.hash_to_field([b"abc"], b"dst", 3),                      # count must be 2 or 4
        lambda: bn254.hash_to_g1(z(10), b"dst"),                              # device-style messages without an offset table
        lambda: bn254.hash_to_g2(z(10), b"dst", msg_off=torch.tensor([0, 10], dtype=torch.int32)),   # table must be int64
        lambda: bn254.hash_to_g1(z(10), b"dst", msg_off=torch.tensor([0, 10], dtype=torch.int64)),   # right shapes, but host tensors
        lambda: bn254.hash_to_field(np.frombuffer(b"abc", dtype=np.uint8), b"dst", 2, msg_off=np.array([0, 5], dtype=np.uint64)),   # offsets beyond the buffer
        lambda: bn254.g1_scalar_mul(np.zeros(64, dtype=np.uint8), np.zeros(4 * 32, dtype=np.uint8), out=np.zeros(3 * 64, dtype=np.uint8)),   # host out too small
    ]
    bn254._slots = bn254._slots or {0: 0}                                     # as after init(0); no device is touched below
    for i, call in enumerate(bad):
        with pytest.raises(ValueError):
            call()
        assert True, i


def test_device_list_and_collective_entries_fail_loudly_without_gpu(lib):
    """The multi-device and RCCL entries of the boundary: without a bound gfx950 device every one of them returns a negative
    status with a message — none aborts, none pretends."""
    import ctypes
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: tests/cpp/test_multi_device.cpp covers the working paths")
    devs = (ctypes.c_int * 2)(0, 1)
    assert lib.gpbc_init_devices(devs, 2) < 0 and b"HIP" in lib.gpbc_last_error() or b"device" in lib.gpbc_last_error()
    assert lib.gpbc_num_devices() == 0
    assert lib.gpbc_set_device(0) < 0 and lib.gpbc_get_device() < 0 and lib.gpbc_device_at(0) < 0
    assert lib.gpbc_comm_init_all() < 0
    assert lib.gpbc_comm_ranks() == 0 and lib.gpbc_comm_rank() < 0
    buf = (ctypes.c_uint8 * 128)()
    assert lib.gpbc_comm_init_rank(buf, 2, 0) < 0                      # no device bound
    assert lib.gpbc_comm_init_rank(buf, 2, 5) < 0                      # rank out of range
    assert lib.gpbc_allgather_dev(buf, 128, buf, None) < 0
    out = np.zeros(64, dtype=np.uint8)
    pts, ks = np.zeros(64 * 4, dtype=np.uint8), np.zeros(32 * 4, dtype=np.uint8)
    assert lib.gpbc_g1_scalar_mul_sum(pts.ctypes.data, ks.ctypes.data, ctypes.c_size_t(4), out.ctypes.data) < 0
    assert lib.gpbc_check_segments_dev(None, ctypes.c_size_t(1), ctypes.c_size_t(1), None) < 0
    assert lib.gpbc_set_host_sharding(1) == 0 and lib.gpbc_comm_destroy() == 0 and lib.gpbc_shutdown() == 0
