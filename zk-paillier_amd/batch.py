"""Structure-of-arrays containers for batches of RangeProofNi proofs (the buffers behind
zkp_range_ni_proofs / zkp_range_ni_witness of include/zkp_hip.h).

Buffers are numpy arrays (host pointers) or torch CUDA tensors (device pointers, resident
in HBM).  torch is used for device memory only."""
import numpy as np

from . import capi

_PROOF_FIELDS = ("n", "range", "ciphertext", "c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2")
_WIT_FIELDS = ("x", "r", "w1", "w2", "r1", "r2")


def _alloc(shape, dtype, device):
    if device is None:
        return np.zeros(shape, dtype=dtype)
    import torch
    # torch has no uint32 arithmetic, but it is only a byte container here: int32 view of the same bits
    tdt = {np.uint32: torch.int32, np.uint8: torch.uint8}[dtype]
    return torch.zeros(shape, dtype=tdt, device=device)


class RangeBatch:
    """B proofs x EF rows, key width n_bits (kw = n_bits/32 limbs).

    shared_key=True keeps ONE n for the batch (n_stride = 0)."""

    def __init__(self, n_bits: int, batch: int, error_factor: int = capi.SECURITY_PARAMETER, shared_key: bool = True,
                 device=None):
        self.n_bits, self.batch, self.ef, self.shared_key, self.device = n_bits, batch, error_factor, shared_key, device
        kw = n_bits // 32
        self.kw = kw
        B, EF = batch, error_factor
        u32, u8 = np.uint32, np.uint8
        self.n = _alloc((1 if shared_key else B, kw), u32, device)
        self.range = _alloc((B, kw), u32, device)
        self.ciphertext = _alloc((B, 2 * kw), u32, device)
        self.c1 = _alloc((B, EF, 2 * kw), u32, device)
        self.c2 = _alloc((B, EF, 2 * kw), u32, device)
        self.resp_kind = _alloc((B, EF), u8, device)
        self.resp_j = _alloc((B, EF), u8, device)
        self.resp_w1 = _alloc((B, EF, kw), u32, device)
        self.resp_r1 = _alloc((B, EF, kw), u32, device)
        self.resp_w2 = _alloc((B, EF, kw), u32, device)
        self.resp_r2 = _alloc((B, EF, kw), u32, device)

    def struct(self) -> capi.RangeNiProofs:
        s = capi.RangeNiProofs()
        s.n_bits, s.error_factor, s.batch = self.n_bits, self.ef, self.batch
        s.n_stride = 0 if self.shared_key else self.kw
        for f in _PROOF_FIELDS:
            setattr(s, f, capi.ptr(getattr(self, f)))
        return s

    def to(self, device):
        """copy to another memory space (None = host numpy)"""
        out = RangeBatch.__new__(RangeBatch)
        out.__dict__.update(self.__dict__)
        out.device = device
        for f in _PROOF_FIELDS:
            setattr(out, f, _move(getattr(self, f), device))
        return out

    def slice(self, lo, hi):
        out = RangeBatch.__new__(RangeBatch)
        out.__dict__.update(self.__dict__)
        out.batch = hi - lo
        for f in _PROOF_FIELDS:
            a = getattr(self, f)
            if f == "n" and self.shared_key:
                setattr(out, f, a)
            else:
                setattr(out, f, a[lo:hi])
        return out


class RangeWitness:
    def __init__(self, n_bits, batch, error_factor=capi.SECURITY_PARAMETER, device=None):
        kw = n_bits // 32
        self.kw, self.batch, self.ef, self.device = kw, batch, error_factor, device
        self.x = _alloc((batch, kw), np.uint32, device)
        self.r = _alloc((batch, kw), np.uint32, device)
        for f in ("w1", "w2", "r1", "r2"):
            setattr(self, f, _alloc((batch, error_factor, kw), np.uint32, device))

    def struct(self) -> capi.RangeNiWitness:
        s = capi.RangeNiWitness()
        for f in _WIT_FIELDS:
            setattr(s, f, capi.ptr(getattr(self, f)))
        return s

    def slice(self, lo, hi):
        out = RangeWitness.__new__(RangeWitness)
        out.__dict__.update(self.__dict__)
        out.batch = hi - lo
        for f in _WIT_FIELDS:
            setattr(out, f, getattr(self, f)[lo:hi])
        return out

    def to(self, device):
        out = RangeWitness.__new__(RangeWitness)
        out.__dict__.update(self.__dict__)
        out.device = device
        for f in _WIT_FIELDS:
            setattr(out, f, _move(getattr(self, f), device))
        return out


def _move(a, device):
    if isinstance(a, np.ndarray):
        if device is None:
            return a.copy()
        import torch
        if a.dtype == np.uint32:
            return torch.from_numpy(a.view(np.int32)).to(device)
        return torch.from_numpy(a).to(device)
    # torch tensor
    if device is None:
        h = a.cpu().numpy()
        return h.view(np.uint32) if h.dtype == np.int32 else h
    return a.to(device)


def make_range_witness(n_bits, batch, error_factor=capi.SECURITY_PARAMETER, device=None):
    return RangeWitness(n_bits, batch, error_factor, device)
