"""Host-side mirror of the gnark-crypto bn254 surface the reference calls, over the C ABI.

Reference call surface (gnark-crypto v0.19.0 `ecc/bn254`, used e.g. at
signature/bls01_signature/bls_signature.go:45,63,81 and cpabe/bsw07/bsw07_cpabe.go:75,184):

    bn254.Pair(P []G1Affine, Q []G2Affine) (GT, error)            -> pair(P, Q)
    bn254.PairingCheck(P, Q) (bool, error)                        -> pairing_check(P, Q)
    (*G1Affine).ScalarMultiplication(a, s) / ...Base(s)           -> g1_scalar_mul(a, s) / g1_scalar_mul_base(s)
    (*G2Affine).ScalarMultiplication(a, s) / ...Base(s)           -> g2_scalar_mul(a, s) / g2_scalar_mul_base(s)
    (*GT).Exp / Mul / Div / Inverse                               -> gt_exp / gt_mul / gt_div / gt_inverse
    bn254.Generators()                                            -> generators()

plus the batched forms the engine adds (pair_batch, multi_pair, pairing_check_batch).  Points and GT
values are numpy uint8 arrays (host) or torch uint8 CUDA tensors (HBM-resident) holding gnark in-memory
structs (Montgomery little-endian limbs): G1 64 B, G2 128 B, GT 384 B; scalars 32-byte little-endian.
Errors follow gnark: length mismatch or empty input to pair/pairing_check raises ValueError("invalid
inputs sizes").  Everything computes on the GPU; a missing extension or device raises EngineError.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import EngineError  # noqa: F401  (re-export)

G1_BYTES, G2_BYTES, GT_BYTES, SCALAR_BYTES = 64, 128, 384, 32
R_ORDER = 21888242871839275222246405745257275088548364400416034343698204186575808495617

# gnark Generators(): g1 = (1, 2), g2 = the standard alt_bn128 twist generator; Montgomery-form bytes.
_G1_GEN_HEX = (
    "9d0d8fc58d435dd33d0bc7f528eb780a2c4679786fa36e662fdf079ac1770a0e3a1b1e8b1b87baa67b168eeb51d6f114"
    "588cf2f0de46ddcc5ebe0f3483ef141c")
_G2_GEN_HEX = (
    "2620bc02d1b5838e72017b493519ebdcdf1a81974726b8fb3b5096af4138571940614ca87d73b4afc4d802585add4360"
    "862fa052fc50e9096b7bea3a83f0fe14f6e96b889dfa9d61789b9ef597d27ffefe7d1b23621a9eff06429eaeeb7efd28"
    "ee5618c7565b0964bb3c7d3222f957dc76103533be35f9558264fd93e6a0a40d")

_slots = None          # HIP ordinal -> slot index of the bound device list (None before init)


def init(device=0):
    """Bind the process to one HIP device (one process per GPU: `init(local_rank)`) or to a list of them (one process
    driving several GPUs: host-pointer batch calls then shard over all of them; CUDA tensors pick their own device)."""
    global _slots
    lib = _lib.load()
    devs = [int(device)] if isinstance(device, int) else [int(d) for d in device]
    arr = (ctypes.c_int * len(devs))(*devs)
    _lib.check(lib.gpbc_init_devices(arr, ctypes.c_int(len(devs))))
    _slots = {}
    for i, d in enumerate(devs):
        _slots.setdefault(d, i)
    return device


def init_all():
    """Bind every visible device; returns their number."""
    n = _lib.check(_lib.load().gpbc_device_count())
    init(list(range(n)))
    return n


def num_devices():
    return int(_lib.load().gpbc_num_devices())


def release_workspaces():
    """Hand the library's grow-only device buffers (workspaces, per-call scratch) back to the driver; they regrow on demand."""
    _ensure_init()
    _lib.check(_lib.load().gpbc_release_workspaces())


def shutdown():
    global _slots
    for t in list(_gen_tables.values()):
        t.close()
    _gen_tables.clear()
    _lib.check(_lib.load().gpbc_shutdown())
    _slots = None


def _ensure_init():
    if _slots is None:
        init(0)


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _np(x, width=None):
    a = np.ascontiguousarray(x, dtype=np.uint8).reshape(-1)
    if width is not None and a.size % width:
        raise ValueError("buffer length %d is not a multiple of %d" % (a.size, width))
    return a


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _sz(n):
    return ctypes.c_size_t(n)


def scalars_to_bytes(scalars):
    """ints (any sign/size; reduced mod r like fr.Element.BigInt round trips) or a uint8 buffer -> n x 32 LE bytes."""
    if isinstance(scalars, (bytes, bytearray, np.ndarray)) or _is_torch(scalars):
        return scalars
    if isinstance(scalars, int):
        scalars = [scalars]
    return np.frombuffer(b"".join((int(s) % R_ORDER).to_bytes(32, "little") for s in scalars), dtype=np.uint8)


def generators():
    """(g1, g2) affine generators as uint8 arrays (gnark: `_, _, g1, g2 := bn254.Generators()`)."""
    return (np.frombuffer(bytes.fromhex(_G1_GEN_HEX), dtype=np.uint8).copy(),
            np.frombuffer(bytes.fromhex(_G2_GEN_HEX), dtype=np.uint8).copy())


P_MODULUS = 21888242871839275222246405745257275088696311157297823662689037894645226208583


def _fp_neg_bytes(b):
    """p - y on one 32-byte little-endian Montgomery value (the negation of yR mod p is (p - y)R mod p); 0 stays 0."""
    v = int.from_bytes(bytes(b), "little")
    return ((P_MODULUS - v) % P_MODULUS).to_bytes(32, "little")


def g1_neg(pt):
    """G1Affine.Neg on one 64-byte point (host side, like the field negation of include/gpbc_bn254.hpp): (x, -y)."""
    b = np.asarray(pt, dtype=np.uint8).reshape(G1_BYTES).tobytes()
    return np.frombuffer(b[:32] + _fp_neg_bytes(b[32:]), dtype=np.uint8).copy()


def g2_neg(pt):
    """G2Affine.Neg on one 128-byte point: (x, -y) with y in Fp2."""
    b = np.asarray(pt, dtype=np.uint8).reshape(G2_BYTES).tobytes()
    return np.frombuffer(b[:64] + _fp_neg_bytes(b[64:96]) + _fp_neg_bytes(b[96:]), dtype=np.uint8).copy()


# --------------------------------------------------------------------------------------- torch (HBM-resident) path
def _torch_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _tptr(t):
    if t.dtype.__str__() != "torch.uint8" or not t.is_cuda or not t.is_contiguous():
        raise ValueError("device buffers must be contiguous uint8 CUDA tensors")
    return ctypes.c_void_p(t.data_ptr())


def _tchk(first, *specs):
    """Validate the HBM-resident arguments of one call before their raw pointers cross the C ABI: every (tensor, bytes,
    name) must be a contiguous uint8 CUDA tensor of exactly `bytes` bytes on the device of `first`; that device must be one
    of the bound ones and becomes the calling thread's current device.  A wrong size here would be an out-of-bounds device
    access inside a kernel, so it is a ValueError on the host instead."""
    dev = first.device
    for t, nbytes, name in specs:
        if not _is_torch(t) or t.dtype.__str__() != "torch.uint8" or not t.is_contiguous():
            raise ValueError("%s must be a contiguous uint8 CUDA tensor" % name)
        if t.numel() != nbytes:
            raise ValueError("%s holds %d bytes, expected %d" % (name, t.numel(), nbytes))
        if t.device != dev:
            raise ValueError("%s is on %s, expected %s" % (name, t.device, dev))
    if not first.is_cuda:
        raise ValueError("device buffers must be CUDA tensors (host data goes in as numpy arrays)")
    idx = dev.index if dev.index is not None else 0
    if _slots is None or idx not in _slots:
        raise ValueError("device %s is not bound: call bn254.init() with it" % dev)
    _lib.check(_lib.load().gpbc_set_device(ctypes.c_int(_slots[idx])))


def _tnew(like, n, width):
    import torch
    return torch.empty((n, width), dtype=torch.uint8, device=like.device)


# --------------------------------------------------------------------------------------- pairings
def pair_batch(P, Q, out=None):
    """n independent pairings: out[i] = Pair([P[i]], [Q[i]])."""
    _ensure_init()
    lib = _lib.load()
    if _is_torch(P):
        n = P.numel() // G1_BYTES
        if not _is_torch(Q) or Q.numel() // G2_BYTES != n or n == 0:
            raise ValueError("invalid inputs sizes")
        out = _tnew(P, n, GT_BYTES) if out is None else out
        _tchk(P, (P, n * G1_BYTES, "P"), (Q, n * G2_BYTES, "Q"), (out, n * GT_BYTES, "out"))
        _lib.check(lib.gpbc_pair_batch_dev(_tptr(P), _tptr(Q), _sz(n), _tptr(out), _torch_stream()))
        return out
    P, Q = _np(P, G1_BYTES), _np(Q, G2_BYTES)
    n = P.size // G1_BYTES
    if Q.size // G2_BYTES != n or n == 0:
        raise ValueError("invalid inputs sizes")
    if out is None:
        out = np.empty((n, GT_BYTES), dtype=np.uint8)
    elif not (isinstance(out, np.ndarray) and out.dtype == np.uint8 and out.size == n * GT_BYTES and out.flags["C_CONTIGUOUS"] and out.flags["WRITEABLE"]):
        raise ValueError("out must be a writable contiguous uint8 array of %d bytes" % (n * GT_BYTES))
    _lib.check(lib.gpbc_pair_batch(_ptr(P), _ptr(Q), _sz(n), _ptr(out)))
    return out


def multi_pair(P, Q, seg_off, out=None, workspace=None):
    """k products of pairings: out[j] = Pair(P[seg_off[j]:seg_off[j+1]], Q[...]) with one final exponentiation each."""
    _ensure_init()
    lib = _lib.load()
    if _is_torch(P):
        import torch
        n = P.numel() // G1_BYTES
        if not _is_torch(Q) or Q.numel() // G2_BYTES != n:
            raise ValueError("invalid inputs sizes")
        if not _is_torch(seg_off):
            # segment table on the host: the engine can cut segments into chunks that share their Miller squarings
            seg = np.ascontiguousarray(seg_off, dtype=np.uint64)
            k = seg.size - 1
            if k < 1 or int(seg[-1]) != n:
                raise ValueError("invalid inputs sizes")
            out = _tnew(P, k, GT_BYTES) if out is None else out
            _tchk(P, (P, n * G1_BYTES, "P"), (Q, n * G2_BYTES, "Q"), (out, k * GT_BYTES, "out"))
            _lib.check(lib.gpbc_multi_pair_hostseg_dev(_tptr(P), _tptr(Q), _ptr(seg), _sz(k), _tptr(out), _torch_stream()))
            return out
        # segment table in device memory: it is read as k+1 uint64 by the kernel, so its dtype and size are checked here
        if seg_off.dtype not in (torch.int64, torch.uint64) or not seg_off.is_cuda or not seg_off.is_contiguous() or seg_off.device != P.device:
            raise ValueError("a device segment table must be a contiguous int64 / uint64 CUDA tensor on the points' device")
        k = seg_off.numel() - 1
        if k < 1:
            raise ValueError("invalid inputs sizes")
        _tchk(P, (P, n * G1_BYTES, "P"))                     # binds the device before the table is checked on it
        if lib.gpbc_check_segments_dev(ctypes.c_void_p(seg_off.data_ptr()), _sz(n), _sz(k), _torch_stream()) < 0:
            raise ValueError("invalid inputs sizes: " + lib.gpbc_last_error().decode())   # must start at 0, be monotone, end at n
        out = _tnew(P, k, GT_BYTES) if out is None else out
        wsb = lib.gpbc_multi_pair_workspace_bytes(n, k)
        if workspace is None:
            workspace = torch.empty(max(wsb, 1), dtype=torch.uint8, device=P.device)
        if workspace.numel() < wsb:
            raise ValueError("workspace holds %d bytes, needs %d" % (workspace.numel(), wsb))
        _tchk(P, (P, n * G1_BYTES, "P"), (Q, n * G2_BYTES, "Q"), (out, k * GT_BYTES, "out"), (workspace, workspace.numel(), "workspace"))
        _lib.check(lib.gpbc_multi_pair_dev(_tptr(P), _tptr(Q), ctypes.c_void_p(seg_off.data_ptr()), _sz(n), _sz(k),
                                           _tptr(out), _tptr(workspace), _sz(workspace.numel()), _torch_stream()))
        return out
    P, Q = _np(P, G1_BYTES), _np(Q, G2_BYTES)
    seg = np.ascontiguousarray(seg_off, dtype=np.uint64)
    k = seg.size - 1
    if Q.size // G2_BYTES != P.size // G1_BYTES or k < 1 or int(seg[-1]) != P.size // G1_BYTES:
        raise ValueError("invalid inputs sizes")
    out = np.empty((k, GT_BYTES), dtype=np.uint8)
    _lib.check(lib.gpbc_multi_pair(_ptr(P), _ptr(Q), _ptr(seg), _sz(k), _ptr(out)))
    return out


def multi_pair_fixed_q(P, Q):
    """k products over ONE shared list of m G2 points: out[j] = Pair(P[j*m:(j+1)*m], Q).  The lines of every Q_i are computed
    once for all k segments (a decryption key against k ciphertexts; gnark: PrecomputeLines / MillerLoopFixedQ)."""
    _ensure_init()
    lib = _lib.load()
    if _is_torch(P):
        if not _is_torch(Q):
            raise ValueError("invalid inputs sizes")
        m = Q.numel() // G2_BYTES
        n = P.numel() // G1_BYTES
        if m < 1 or n < m or n % m:
            raise ValueError("invalid inputs sizes")
        out = _tnew(P, n // m, GT_BYTES)
        _tchk(P, (P, n * G1_BYTES, "P"), (Q, m * G2_BYTES, "Q"))
        _lib.check(lib.gpbc_multi_pair_fixed_q_dev(_tptr(P), _tptr(Q), _sz(m), _sz(n // m), _tptr(out), _torch_stream()))
        return out
    P, Q = _np(P, G1_BYTES), _np(Q, G2_BYTES)
    m, n = Q.size // G2_BYTES, P.size // G1_BYTES
    if m < 1 or n < m or n % m:
        raise ValueError("invalid inputs sizes")
    out = np.empty((n // m, GT_BYTES), dtype=np.uint8)
    _lib.check(lib.gpbc_multi_pair_fixed_q(_ptr(P), _ptr(Q), _sz(m), _sz(n // m), _ptr(out)))
    return out


def pair(P, Q):
    """bn254.Pair(P, Q): the product of the pairings of all (P[i], Q[i]); one 384-byte GT."""
    P, Q = _np(P, G1_BYTES), _np(Q, G2_BYTES)
    n = P.size // G1_BYTES
    if n == 0 or Q.size // G2_BYTES != n:
        raise ValueError("invalid inputs sizes")
    return multi_pair(P, Q, [0, n])[0]


def pairing_check_batch(P, Q, seg_off):
    _ensure_init()
    lib = _lib.load()
    P, Q = _np(P, G1_BYTES), _np(Q, G2_BYTES)
    seg = np.ascontiguousarray(seg_off, dtype=np.uint64)
    k = seg.size - 1
    if Q.size // G2_BYTES != P.size // G1_BYTES or k < 1 or int(seg[-1]) != P.size // G1_BYTES:
        raise ValueError("invalid inputs sizes")
    ok = np.empty(k, dtype=np.uint8)
    _lib.check(lib.gpbc_pairing_check(_ptr(P), _ptr(Q), _ptr(seg), _sz(k), _ptr(ok)))
    return ok.astype(bool)


def pairing_check(P, Q):
    """bn254.PairingCheck(P, Q): product of pairings == 1."""
    P, Q = _np(P, G1_BYTES), _np(Q, G2_BYTES)
    n = P.size // G1_BYTES
    if n == 0 or Q.size // G2_BYTES != n:
        raise ValueError("invalid inputs sizes")
    return bool(pairing_check_batch(P, Q, [0, n])[0])


def miller_loop(P, Q):
    _ensure_init()
    lib = _lib.load()
    if _is_torch(P):
        n = P.numel() // G1_BYTES
        out = _tnew(P, n, GT_BYTES)
        _tchk(P, (P, n * G1_BYTES, "P"), (Q, n * G2_BYTES, "Q"))
        _lib.check(lib.gpbc_miller_loop_dev(_tptr(P), _tptr(Q), _sz(n), _tptr(out), _torch_stream()))
        return out
    P, Q = _np(P, G1_BYTES), _np(Q, G2_BYTES)
    n = P.size // G1_BYTES
    out = np.empty((n, GT_BYTES), dtype=np.uint8)
    _lib.check(lib.gpbc_miller_loop(_ptr(P), _ptr(Q), _sz(n), _ptr(out)))
    return out


def final_exp(F):
    _ensure_init()
    lib = _lib.load()
    if _is_torch(F):
        n = F.numel() // GT_BYTES
        out = _tnew(F, n, GT_BYTES)
        _tchk(F, (F, n * GT_BYTES, "F"))
        _lib.check(lib.gpbc_final_exp_dev(_tptr(F), _sz(n), _tptr(out), _torch_stream()))
        return out
    F = _np(F, GT_BYTES)
    n = F.size // GT_BYTES
    out = np.empty((n, GT_BYTES), dtype=np.uint8)
    _lib.check(lib.gpbc_final_exp(_ptr(F), _sz(n), _ptr(out)))
    return out


# --------------------------------------------------------------------------------------- scalar multiplication
def _scalar_mul(width, host_fn, dev_fn, bases, scalars, out):
    _ensure_init()
    scalars = scalars_to_bytes(scalars)
    if _is_torch(scalars):
        if not _is_torch(bases):
            raise ValueError("bases and scalars must both be CUDA tensors (or both host buffers)")
        n = scalars.numel() // SCALAR_BYTES
        nbase = bases.numel() // width
        if nbase not in (1, n):
            raise ValueError("need one base or one base per scalar")
        out = _tnew(scalars, n, width) if out is None else out
        _tchk(scalars, (scalars, n * SCALAR_BYTES, "scalars"), (bases, nbase * width, "bases"), (out, n * width, "out"))
        _lib.check(dev_fn(_tptr(bases), _sz(nbase), _tptr(scalars), _sz(n), _tptr(out), _torch_stream()))
        return out
    bases, scalars = _np(bases, width), _np(scalars, SCALAR_BYTES)
    n, nbase = scalars.size // SCALAR_BYTES, bases.size // width
    if nbase not in (1, n):
        raise ValueError("need one base or one base per scalar")
    if out is None:
        out = np.empty((n, width), dtype=np.uint8)
    elif not (isinstance(out, np.ndarray) and out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and out.flags["WRITEABLE"] and out.size == n * width):
        raise ValueError("out must be a writable contiguous uint8 array of %d bytes" % (n * width))
    _lib.check(host_fn(_ptr(bases), _sz(nbase), _ptr(scalars), _sz(n), _ptr(out)))
    return out


def g1_scalar_mul(bases, scalars, out=None):
    """out[i] = new(G1Affine).ScalarMultiplication(&bases[i], scalars[i]) (one shared base allowed)."""
    lib = _lib.load()
    return _scalar_mul(G1_BYTES, lib.gpbc_g1_scalar_mul_batch, lib.gpbc_g1_scalar_mul_batch_dev, bases, scalars, out)


def g2_scalar_mul(bases, scalars, out=None):
    lib = _lib.load()
    return _scalar_mul(G2_BYTES, lib.gpbc_g2_scalar_mul_batch, lib.gpbc_g2_scalar_mul_batch_dev, bases, scalars, out)


_gen_tables = {}


def _scalar_mul_base(g2, scalars):
    """Small host-side calls go through fixed-base window tables of the generator (built on the first call, kept until shutdown():
    32 mixed additions per multiplication instead of the variable-base kernel's doublings — what include/gpbc_bn254.hpp and the Go
    shim do); large batches and device tensors take the shared-base form of the variable-base kernel, which builds its own table."""
    k = scalars_to_bytes(scalars)
    if not _is_torch(k):
        k = _np(k, SCALAR_BYTES)
    if _is_torch(k) or k.size // SCALAR_BYTES >= 16384:
        return (g2_scalar_mul if g2 else g1_scalar_mul)(generators()[1 if g2 else 0], k)
    _ensure_init()
    if g2 not in _gen_tables:
        _gen_tables[g2] = FixedBase(generators()[1 if g2 else 0], g2=g2)
    return _gen_tables[g2].mul(k)


def g1_scalar_mul_base(scalars):
    """ScalarMultiplicationBase: [s]g1."""
    return _scalar_mul_base(False, scalars)


def g2_scalar_mul_base(scalars):
    return _scalar_mul_base(True, scalars)


def _sum(width, is_g2, host_fn, dev_fn, pts):
    _ensure_init()
    lib = _lib.load()
    if _is_torch(pts):
        import torch
        n = pts.numel() // width
        out = _tnew(pts, 1, width)
        wsb = lib.gpbc_sum_workspace_bytes(n, is_g2)
        ws = torch.empty(max(wsb, 1), dtype=torch.uint8, device=pts.device)
        _tchk(pts, (pts, n * width, "points"))
        _lib.check(dev_fn(_tptr(pts), _sz(n), _tptr(out), _tptr(ws), _sz(ws.numel()), _torch_stream()))
        return out[0]
    pts = _np(pts, width)
    out = np.empty(width, dtype=np.uint8)
    _lib.check(host_fn(_ptr(pts), _sz(pts.size // width), _ptr(out)))
    return out


def g1_sum(pts):
    """Sum of affine G1 points (chains of G1Affine.Add in the reference)."""
    lib = _lib.load()
    return _sum(G1_BYTES, 0, lib.gpbc_g1_sum, lib.gpbc_g1_sum_dev, pts)


def g2_sum(pts):
    lib = _lib.load()
    return _sum(G2_BYTES, 1, lib.gpbc_g2_sum, lib.gpbc_g2_sum_dev, pts)


def _scalar_mul_sum(width, host_fn, dev_fn, bases, scalars):
    _ensure_init()
    scalars = scalars_to_bytes(scalars)
    if _is_torch(scalars):
        if not _is_torch(bases):
            raise ValueError("bases and scalars must both be CUDA tensors (or both host buffers)")
        n = scalars.numel() // SCALAR_BYTES
        out = _tnew(scalars, 1, width)
        _tchk(scalars, (scalars, n * SCALAR_BYTES, "scalars"), (bases, n * width, "bases"))
        _lib.check(dev_fn(_tptr(bases), _tptr(scalars), _sz(n), _tptr(out), _torch_stream()))
        return out[0]
    bases, scalars = _np(bases, width), _np(scalars, SCALAR_BYTES)
    n = scalars.size // SCALAR_BYTES
    if bases.size // width != n:
        raise ValueError("need one base per scalar")
    out = np.empty(width, dtype=np.uint8)
    _lib.check(host_fn(_ptr(bases), _ptr(scalars), _sz(n), _ptr(out)))
    return out


def g1_scalar_mul_sum(bases, scalars):
    """sum_i [s_i] P_i (the verifier's sums of BLS aggregate verification, BASELINE config 3).  Host buffers: sharded over the
    bound devices.  CUDA tensors: this rank's shard; with a communicator (comm_init_rank) the result is the sum over ALL
    ranks — one RCCL all-gather of a point per rank inside the library."""
    lib = _lib.load()
    return _scalar_mul_sum(G1_BYTES, lib.gpbc_g1_scalar_mul_sum, lib.gpbc_g1_scalar_mul_sum_dev, bases, scalars)


def g2_scalar_mul_sum(bases, scalars):
    lib = _lib.load()
    return _scalar_mul_sum(G2_BYTES, lib.gpbc_g2_scalar_mul_sum, lib.gpbc_g2_scalar_mul_sum_dev, bases, scalars)


# --------------------------------------------------------------------------------------- collectives (RCCL inside the library)
COMM_ID_BYTES = 128


def comm_init_all():
    """One RCCL communicator over all bound devices of this process (rank = device slot)."""
    _ensure_init()
    _lib.check(_lib.load().gpbc_comm_init_all())


def comm_unique_id():
    """128 opaque bytes from rank 0, to be handed to every rank's comm_init_rank (any transport: torch.distributed
    broadcast, a file, the launcher's environment)."""
    buf = (ctypes.c_uint8 * COMM_ID_BYTES)()
    _lib.check(_lib.load().gpbc_comm_get_unique_id(buf))
    return bytes(buf)


def comm_init_rank(unique_id, n_ranks, rank):
    """Join the multi-process communicator as `rank` with this process's current device (one process per GPU)."""
    _ensure_init()
    if len(unique_id) != COMM_ID_BYTES:
        raise ValueError("the communicator id is %d bytes" % COMM_ID_BYTES)
    buf = (ctypes.c_uint8 * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
    _lib.check(_lib.load().gpbc_comm_init_rank(buf, ctypes.c_int(n_ranks), ctypes.c_int(rank)))


def comm_ranks():
    return int(_lib.load().gpbc_comm_ranks())


def comm_destroy():
    _lib.check(_lib.load().gpbc_comm_destroy())


def allgather(send, out=None):
    """All-gather equal-sized uint8 CUDA blocks over the library's communicator: returns [n_ranks, send.numel()].
    Enqueued on the current torch stream, not synchronised."""
    _ensure_init()
    import torch
    ranks = comm_ranks()
    if ranks < 1:
        raise EngineError("no communicator: call comm_init_rank() / comm_init_all() first")
    nb = send.numel()
    out = torch.empty((ranks, nb), dtype=torch.uint8, device=send.device) if out is None else out
    _tchk(send, (send, nb, "send"), (out, ranks * nb, "out"))
    _lib.check(_lib.load().gpbc_allgather_dev(_tptr(send), _sz(nb), _tptr(out), _torch_stream()))
    return out


# --------------------------------------------------------------------------------------- GT
def gt_exp(x, k, out=None):
    """out[i] = new(GT).Exp(x[i], k[i]); Python ints may be negative (inverse, as gnark)."""
    _ensure_init()
    lib = _lib.load()
    if isinstance(k, int):
        k = [k]
    if isinstance(k, (list, tuple)) and any(int(s) < 0 for s in k):
        neg = np.array([int(s) < 0 for s in k])
        xs = _np(x, GT_BYTES).reshape(-1, GT_BYTES).copy()
        xs[neg] = gt_inverse(xs[neg])
        x = xs
        k = np.frombuffer(b"".join(abs(int(s)).to_bytes(32, "little") for s in k), dtype=np.uint8)
    elif isinstance(k, (list, tuple)):
        k = np.frombuffer(b"".join(int(s).to_bytes(32, "little") for s in k), dtype=np.uint8)
    if _is_torch(x):
        n = x.numel() // GT_BYTES
        if not _is_torch(k):
            import torch
            k = torch.from_numpy(np.ascontiguousarray(_np(k, SCALAR_BYTES)).copy()).to(x.device)
        out = _tnew(x, n, GT_BYTES) if out is None else out
        _tchk(x, (x, n * GT_BYTES, "x"), (k, n * SCALAR_BYTES, "k"), (out, n * GT_BYTES, "out"))
        _lib.check(lib.gpbc_gt_exp_batch_dev(_tptr(x), _tptr(k), _sz(n), _tptr(out), _torch_stream()))
        return out
    x, k = _np(x, GT_BYTES), _np(k, SCALAR_BYTES)
    n = x.size // GT_BYTES
    if k.size // SCALAR_BYTES != n:
        raise ValueError("one exponent per element")
    out = np.empty((n, GT_BYTES), dtype=np.uint8)
    _lib.check(lib.gpbc_gt_exp_batch(_ptr(x), _ptr(k), _sz(n), _ptr(out)))
    return out


def _gt_binary(host_fn, dev_fn, a, b):
    _ensure_init()
    if _is_torch(a):
        n = a.numel() // GT_BYTES
        out = _tnew(a, n, GT_BYTES)
        _tchk(a, (a, n * GT_BYTES, "a"), (b, n * GT_BYTES, "b"))
        _lib.check(dev_fn(_tptr(a), _tptr(b), _sz(n), _tptr(out), _torch_stream()))
        return out
    a, b = _np(a, GT_BYTES), _np(b, GT_BYTES)
    n = a.size // GT_BYTES
    if b.size != a.size:
        raise ValueError("operand sizes differ")
    out = np.empty((n, GT_BYTES), dtype=np.uint8)
    _lib.check(host_fn(_ptr(a), _ptr(b), _sz(n), _ptr(out)))
    return out


def gt_mul(a, b):
    lib = _lib.load()
    return _gt_binary(lib.gpbc_gt_mul_batch, lib.gpbc_gt_mul_batch_dev, a, b)


def gt_div(a, b):
    lib = _lib.load()
    return _gt_binary(lib.gpbc_gt_div_batch, lib.gpbc_gt_div_batch_dev, a, b)


def gt_inverse(a):
    _ensure_init()
    lib = _lib.load()
    if _is_torch(a):
        n = a.numel() // GT_BYTES
        out = _tnew(a, n, GT_BYTES)
        _tchk(a, (a, n * GT_BYTES, "a"))
        _lib.check(lib.gpbc_gt_inverse_batch_dev(_tptr(a), _sz(n), _tptr(out), _torch_stream()))
        return out
    a = _np(a, GT_BYTES)
    n = a.size // GT_BYTES
    out = np.empty((n, GT_BYTES), dtype=np.uint8)
    _lib.check(lib.gpbc_gt_inverse_batch(_ptr(a), _sz(n), _ptr(out)))
    return out


def fp_mul(a, b):
    """Batched Fp Montgomery product (kernel unit test entry)."""
    _ensure_init()
    lib = _lib.load()
    a, b = _np(a, 32), _np(b, 32)
    n = a.size // 32
    out = np.empty((n, 32), dtype=np.uint8)
    _lib.check(lib.gpbc_fp_mul_batch(_ptr(a), _ptr(b), _sz(n), _ptr(out)))
    return out


# --------------------------------------------------------------------------------------- wire formats
# gnark Marshal()/RawBytes(), Bytes() and Unmarshal()/SetBytes() (reference serialization/serialization_curve.go:5-33,
# ibe/gentry06_ibe/gentry06_ibe.go:322-324, hash/hash_from_gt.go:5-8), batched.  Encodings are rows of
# 64 / 32 (G1), 128 / 64 (G2) and 384 (GT) big-endian canonical bytes.
_WIRE = {"g1": (G1_BYTES, 64, 32), "g2": (G2_BYTES, 128, 64), "gt": (GT_BYTES, 384, 384)}


def _marshal(kind, x, compressed):
    _ensure_init()
    lib = _lib.load()
    mem, raw, comp = _WIRE[kind]
    width = comp if compressed else raw
    host = getattr(lib, "gpbc_%s_marshal_batch" % kind)
    dev = getattr(lib, "gpbc_%s_marshal_batch_dev" % kind)
    cflag = () if kind == "gt" else (ctypes.c_int(1 if compressed else 0),)
    if _is_torch(x):
        n = x.numel() // mem
        out = _tnew(x, n, width)
        _tchk(x, (x, n * mem, kind + " elements"))
        _lib.check(dev(_tptr(x), _sz(n), *cflag, _tptr(out), _torch_stream()))
        return out
    x = _np(x, mem)
    n = x.size // mem
    out = np.empty((n, width), dtype=np.uint8)
    _lib.check(host(_ptr(x), _sz(n), *cflag, _ptr(out)))
    return out


def _unmarshal(kind, buf, elem_bytes):
    _ensure_init()
    lib = _lib.load()
    mem, raw, comp = _WIRE[kind]
    if elem_bytes is None:
        elem_bytes = raw
    if elem_bytes not in (raw, comp):
        raise ValueError("%s element size must be %d or %d" % (kind, comp, raw))
    host = getattr(lib, "gpbc_%s_unmarshal_batch" % kind)
    dev = getattr(lib, "gpbc_%s_unmarshal_batch_dev" % kind)
    eb = () if kind == "gt" else (_sz(elem_bytes),)
    if _is_torch(buf):
        import torch
        if buf.numel() % elem_bytes:
            raise ValueError("buffer length %d is not a multiple of %d" % (buf.numel(), elem_bytes))
        n = buf.numel() // elem_bytes
        out = _tnew(buf, n, mem)
        ok = torch.empty((n,), dtype=torch.uint8, device=buf.device)
        _tchk(buf, (buf, n * elem_bytes, kind + " encodings"))
        _lib.check(dev(_tptr(buf), *eb, _sz(n), _tptr(out), _tptr(ok), _torch_stream()))
        return out, ok
    buf = _np(buf, elem_bytes)
    n = buf.size // elem_bytes
    out = np.empty((n, mem), dtype=np.uint8)
    ok = np.empty((n,), dtype=np.uint8)
    _lib.check(host(_ptr(buf), *eb, _sz(n), _ptr(out), _ptr(ok)))
    return out, ok


def g1_marshal(pts, compressed=False):
    """G1Affine.Marshal() (64 B rows) or, compressed, G1Affine.Bytes() (32 B rows)."""
    return _marshal("g1", pts, compressed)


def g2_marshal(pts, compressed=False):
    """G2Affine.Marshal() (128 B rows) or, compressed, G2Affine.Bytes() (64 B rows)."""
    return _marshal("g2", pts, compressed)


def gt_marshal(gt):
    """GT.Marshal() = GT.Bytes(): 384 B rows, coefficients C1.B2.A1 ... C0.B0.A0."""
    return _marshal("gt", gt, False)


def g1_unmarshal(buf, elem_bytes=None):
    """G1Affine.Unmarshal() on rows of `elem_bytes` (64 default, or 32): (points, ok).  ok[i] = 0 where gnark returns an
    error (the point row is then zero); the reference drops that error, callers here should look at it."""
    return _unmarshal("g1", buf, elem_bytes)


def g2_unmarshal(buf, elem_bytes=None):
    """G2Affine.Unmarshal() on rows of `elem_bytes` (128 default, or 64): (points, ok); includes the subgroup check."""
    return _unmarshal("g2", buf, elem_bytes)


def gt_unmarshal(buf):
    """GT.Unmarshal(): (values, ok)."""
    return _unmarshal("gt", buf, None)


# --------------------------------------------------------------------------------------- hash to curve, group part
def _map_fields(width, host_fn, dev_fn, u):
    _ensure_init()
    lib = _lib.load()
    if _is_torch(u):
        n = u.numel() // width
        out = _tnew(u, n, width)
        _tchk(u, (u, n * width, "field elements"))
        _lib.check(dev_fn(_tptr(u), _sz(n), _tptr(out), _torch_stream()))
        return out
    u = _np(u, width)
    n = u.size // width
    out = np.empty((n, width), dtype=np.uint8)
    _lib.check(host_fn(_ptr(u), _sz(n), _ptr(out)))
    return out


def map_to_g1(u):
    """Tail of bn254.HashToG1: rows of two fp.Element (64 B, gnark layout) -> MapToCurve1(u0) + MapToCurve1(u1)."""
    lib = _lib.load()
    return _map_fields(G1_BYTES, lib.gpbc_g1_map_to_curve_batch, lib.gpbc_g1_map_to_curve_batch_dev, u)


def map_to_g2(u):
    """Tail of bn254.HashToG2: rows of two E2 (128 B) -> ClearCofactor(MapToCurve2(u0) + MapToCurve2(u1))."""
    lib = _lib.load()
    return _map_fields(G2_BYTES, lib.gpbc_g2_map_to_curve_batch, lib.gpbc_g2_map_to_curve_batch_dev, u)


def _hash_messages(what, msgs, dst, msg_off=None):
    """Shared body of hash_to_g1 / hash_to_g2 / hash_to_field.  what: 0 G1, 1 G2, 2 / 4 field elements per message.
    msgs: a list of bytes-like messages (host), or the concatenated bytes as a numpy array / CUDA uint8 tensor with
    msg_off (n + 1 offsets; numpy uint64 for host data, an int64 CUDA tensor for device data)."""
    _ensure_init()
    lib = _lib.load()
    dst = bytes(dst)
    if len(dst) > 255:                                           # gnark's ExpandMsgXmd refuses it ("invalid domain size"), so does the Go shim
        raise ValueError("invalid domain size (>255 bytes)")
    width = G1_BYTES if what == 0 else G2_BYTES if what == 1 else 32 * what
    dbuf = ctypes.create_string_buffer(dst, len(dst) if dst else 1)
    count = (ctypes.c_int(what),) if what >= 2 else ()
    if _is_torch(msgs):
        import torch
        if msg_off is None or not _is_torch(msg_off) or msg_off.dtype != torch.int64 or not msg_off.is_contiguous() or msg_off.device != msgs.device:
            raise ValueError("device messages need msg_off as a contiguous int64 tensor on the same device")
        n = msg_off.numel() - 1
        if n < 0:
            raise ValueError("msg_off needs n + 1 entries")
        out = _tnew(msgs, n, width)
        _tchk(msgs, (msgs, msgs.numel(), "messages"))
        fn = (lib.gpbc_hash_to_g1_dev, lib.gpbc_hash_to_g2_dev, lib.gpbc_hash_to_field_dev)[min(what, 2)]
        _lib.check(fn(_tptr(msgs), ctypes.c_void_p(msg_off.data_ptr()), _sz(msgs.numel()), _sz(n), dbuf, _sz(len(dst)), *count, _tptr(out), _torch_stream()))
        return out
    if msg_off is None:
        msgs = [bytes(m) for m in msgs]
        msg_off = np.zeros(len(msgs) + 1, dtype=np.uint64)
        if msgs:
            msg_off[1:] = np.cumsum([len(m) for m in msgs], dtype=np.uint64)
        data = np.frombuffer(b"".join(msgs), dtype=np.uint8) if msgs and int(msg_off[-1]) else np.zeros(1, dtype=np.uint8)
    else:
        data = np.ascontiguousarray(np.asarray(msgs, dtype=np.uint8)).reshape(-1)
        msg_off = np.ascontiguousarray(np.asarray(msg_off, dtype=np.uint64))
        if msg_off.size < 1 or (msg_off.size > 1 and int(msg_off[-1]) > data.size):
            raise ValueError("message offsets exceed the message buffer")
        if data.size == 0:
            data = np.zeros(1, dtype=np.uint8)
    n = msg_off.size - 1
    out = np.empty((n, width), dtype=np.uint8)
    if n:
        fn = (lib.gpbc_hash_to_g1, lib.gpbc_hash_to_g2, lib.gpbc_hash_to_field)[min(what, 2)]
        _lib.check(fn(_ptr(data), _ptr(msg_off), _sz(n), dbuf, _sz(len(dst)), *count, _ptr(out)))
    return out


def hash_to_g1(msgs, dst, msg_off=None):
    """bn254.HashToG1(msg, dst) for every message, hashing included (expand_message_xmd with SHA-256 on the device)."""
    return _hash_messages(0, msgs, dst, msg_off)


def hash_to_g2(msgs, dst, msg_off=None):
    """bn254.HashToG2(msg, dst) for every message, hashing included."""
    return _hash_messages(1, msgs, dst, msg_off)


def hash_to_field(msgs, dst, count=2, msg_off=None):
    """fp.Hash(msg, dst, count) for every message: [n, count * 32] fp.Elements in gnark's layout (count = 2 or 4)."""
    if count not in (2, 4):
        raise ValueError("count must be 2 or 4")
    return _hash_messages(count, msgs, dst, msg_off)


# --------------------------------------------------------------------------------------- fixed-base tables / MSM
class FixedBase:
    """8-bit window tables of a fixed set of bases kept in HBM (1 MB per G1 base, 2 MB per G2 base): afterwards a term of
    a sum costs 32 mixed additions and no doublings.  Serves (*G1Affine|*G2Affine).ScalarMultiplicationBase (one base: the
    generator) and the commitment loops  sum_j [c_j] srs_j  of bibe/afp25_bibe/afp25_bibe_utils.go:44-55.

        fb = FixedBase(bases, g2=False)          # bases: [nbase, 64|128] uint8 (numpy or CUDA tensor)
        out = fb.msm(scalars)                    # scalars: [n_msm, nbase] ints / [n_msm * nbase, 32] bytes -> [n_msm, 64|128]
        out = fb.mul(scalars)                    # nbase == 1: out[i] = [scalars[i]] base
    """

    def __init__(self, bases, g2=False):
        _ensure_init()
        self._lib = _lib.load()
        self.g2 = bool(g2)
        self.width = G2_BYTES if g2 else G1_BYTES
        self._h = ctypes.c_void_p()
        if _is_torch(bases):
            self.nbase = bases.numel() // self.width
            _tchk(bases, (bases, self.nbase * self.width, "bases"))
            _lib.check(self._lib.gpbc_fixed_base_create_dev(ctypes.c_int(1 if g2 else 0), _tptr(bases), _sz(self.nbase),
                                                            _torch_stream(), ctypes.byref(self._h)))
            import torch
            torch.cuda.current_stream().synchronize()            # `bases` may be released by the caller after this returns
        else:
            b = _np(bases, self.width)
            self.nbase = b.size // self.width
            fn = self._lib.gpbc_g2_fixed_base_create if g2 else self._lib.gpbc_g1_fixed_base_create
            _lib.check(fn(_ptr(b), _sz(self.nbase), ctypes.byref(self._h)))

    def table_bytes(self):
        return int(self._lib.gpbc_fixed_base_table_bytes(_sz(self.nbase), ctypes.c_int(1 if self.g2 else 0)))

    def msm(self, scalars):
        k = scalars_to_bytes(scalars) if not (isinstance(scalars, list) and scalars and isinstance(scalars[0], (list, tuple))) \
            else scalars_to_bytes([s for row in scalars for s in row])
        if _is_torch(k):
            import torch
            if k.numel() % (SCALAR_BYTES * self.nbase):
                raise ValueError("need nbase = %d scalars per sum" % self.nbase)
            n = k.numel() // (SCALAR_BYTES * self.nbase)
            out = _tnew(k, n, self.width)
            _tchk(k, (k, n * self.nbase * SCALAR_BYTES, "scalars"))
            wsb = int(self._lib.gpbc_fixed_base_msm_workspace_bytes(self._h, _sz(n)))
            ws = torch.empty((max(wsb, 1),), dtype=torch.uint8, device=k.device)
            _lib.check(self._lib.gpbc_fixed_base_msm_dev(self._h, _tptr(k), _sz(n), _tptr(out), _tptr(ws), _sz(wsb), _torch_stream()))
            torch.cuda.current_stream().synchronize()            # the workspace is freed on return
            return out
        k = _np(k, SCALAR_BYTES)
        if (k.size // SCALAR_BYTES) % self.nbase:
            raise ValueError("need nbase = %d scalars per sum" % self.nbase)
        n = k.size // (SCALAR_BYTES * self.nbase)
        out = np.empty((n, self.width), dtype=np.uint8)
        _lib.check(self._lib.gpbc_fixed_base_msm(self._h, _ptr(k), _sz(n), _ptr(out)))
        return out

    def mul(self, scalars):
        if self.nbase != 1:
            raise ValueError("mul() is the single-base form; use msm()")
        return self.msm(scalars)

    def close(self):
        if self._h:
            self._lib.gpbc_fixed_base_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
