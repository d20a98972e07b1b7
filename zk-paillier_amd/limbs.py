"""Fixed-width little-endian 32-bit limb arrays <-> Python ints / reference byte form.

The C ABI (include/zkp_hip.h) speaks fixed-width limb arrays; the reference speaks
arbitrary-precision BigInt whose byte form is minimal-length big-endian
([upstream] curv BigInt::to_bytes; used by src/zkproofs/utils.rs:15-18)."""
import numpy as np


def int_to_limbs(x: int, nlimbs: int) -> np.ndarray:
    if x < 0 or x.bit_length() > 32 * nlimbs:
        raise ValueError(f"value of {x.bit_length()} bits does not fit {nlimbs} limbs")
    return np.frombuffer(x.to_bytes(4 * nlimbs, "little"), dtype="<u4").copy()


def limbs_to_int(a) -> int:
    return int.from_bytes(np.ascontiguousarray(a, dtype="<u4").tobytes(), "little")


def ints_to_limbs(xs, nlimbs: int) -> np.ndarray:
    """list (or nested list) of ints -> array [..., nlimbs] of uint32."""
    if isinstance(xs, (list, tuple)) and xs and isinstance(xs[0], (list, tuple)):
        return np.stack([ints_to_limbs(v, nlimbs) for v in xs])
    out = np.zeros((len(xs), nlimbs), dtype=np.uint32)
    for i, v in enumerate(xs):
        out[i] = int_to_limbs(v, nlimbs)
    return out


def limbs_to_ints(a):
    a = np.asarray(a)
    if a.ndim == 1:
        return limbs_to_int(a)
    return [limbs_to_ints(r) for r in a]


def to_bytes(x: int) -> bytes:
    """Reference byte form: minimal big-endian, zero -> b'\\x00'."""
    return b"\x00" if x == 0 else x.to_bytes((x.bit_length() + 7) // 8, "big")


def from_bytes(b: bytes) -> int:
    return int.from_bytes(b, "big")
