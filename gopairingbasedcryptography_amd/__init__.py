"""MI355X-native batched BN254 pairing engine: the hot path underneath
mmsyan/GoPairingBasedCryptography (bn254.Pair / G1,G2 ScalarMultiplication / GT ops of gnark-crypto),
as hand-written HIP for gfx950 behind a C ABI (include/gpbc_bn254.h).

    from gopairingbasedcryptography_amd import bn254
    bn254.init(0); gt = bn254.pair_batch(P, Q)
"""
from . import bn254  # noqa: F401
from ._lib import EngineError, LIB_PATH  # noqa: F401

__all__ = ["bn254", "EngineError", "LIB_PATH"]
