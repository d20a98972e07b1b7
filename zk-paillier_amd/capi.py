"""ctypes binding of libzkp_hip.so (include/zkp_hip.h).

There is no CPU fallback: importing this module without the built library, or creating a
context without a gfx950 GPU, raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZKP_HIP_LIB") or os.path.join(_HERE, "libzkp_hip.so")   # (the override is for A/B measurements of kernel variants)

ZKP_OK, ZKP_EINVAL, ZKP_ENONCANONICAL, ZKP_EDEVICE, ZKP_ENOMEM = range(5)
ZKP_F_DEVICE_PTRS = 1
VERDICT_REJECT, VERDICT_ACCEPT, VERDICT_MALFORMED = 0, 1, 2
INV_OK, INV_NONE, INV_DOMAIN = 0, 1, 2
RESP_OPEN, RESP_MASK = 0, 1
SECURITY_PARAMETER = 128
CORRECT_KEY_M2 = 11

u32p = C.POINTER(C.c_uint32)
u8p = C.POINTER(C.c_uint8)


class RangeNiProofs(C.Structure):
    """zkp_range_ni_proofs"""
    _fields_ = [
        ("n_bits", C.c_uint32), ("error_factor", C.c_uint32), ("batch", C.c_uint64),
        ("n_stride", C.c_uint64), ("n", C.c_void_p), ("range", C.c_void_p),
        ("ciphertext", C.c_void_p), ("c1", C.c_void_p), ("c2", C.c_void_p),
        ("resp_kind", C.c_void_p), ("resp_j", C.c_void_p), ("resp_w1", C.c_void_p),
        ("resp_r1", C.c_void_p), ("resp_w2", C.c_void_p), ("resp_r2", C.c_void_p),
    ]


class RangeNiWitness(C.Structure):
    """zkp_range_ni_witness"""
    _fields_ = [("x", C.c_void_p), ("r", C.c_void_p), ("w1", C.c_void_p), ("w2", C.c_void_p),
                ("r1", C.c_void_p), ("r2", C.c_void_p)]


# include/zkp_hip_diag.h: measurement tooling, not the boundary
DIAG_EXPORTS = {
    "zkp_diag_table_traffic": (C.c_int32, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]),
    "zkp_diag_basen": (C.c_int32, [C.c_void_p, C.c_uint32] + [C.c_void_p, C.c_int32] + [C.c_void_p] * 5),
    "zkp_diag_basen_last": (C.c_int32, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint32)]),
    "zkp_diag_basen_engine": (C.c_int32, []),
    "zkp_diag_set_enc_form": (C.c_int32, [C.c_void_p, C.c_int32]),
    "zkp_diag_enc_form": (C.c_int32, [C.c_void_p]),
    "zkp_diag_last_host_blocks": (C.c_int32, [C.c_void_p]),
    "zkp_diag_set_r2l": (C.c_int32, [C.c_void_p, C.c_int32]),
    "zkp_diag_r2l_last": (C.c_int32, [C.c_void_p]),
    "zkp_diag_set_r2l_lanes": (C.c_int32, [C.c_void_p, C.c_int32]),
    "zkp_diag_r2l_lanes_last": (C.c_int32, [C.c_void_p]),
    "zkp_diag_mid_limbs_per_lane": (C.c_int32, [C.c_void_p]),
    "zkp_diag_set_fuse_hash": (C.c_int32, [C.c_void_p, C.c_int32]),
    "zkp_diag_last_fused_hash": (C.c_int32, [C.c_void_p]),
    "zkp_diag_set_split": (C.c_int32, [C.c_void_p, C.c_int32]),
    "zkp_diag_last_split": (C.c_int32, [C.c_void_p]),
    "zkp_diag_set_key_cache": (C.c_int32, [C.c_void_p, C.c_int32]),
    "zkp_diag_key_cache_state": (C.c_int32, [C.c_void_p, C.c_int32, C.POINTER(C.c_uint32)]),
}
ENC_FORM_AUTO, ENC_FORM_N2, ENC_FORM_SHARED, ENC_FORM_ALWAYS = 0, 1, 2, 3
ENC_FORMS = {"auto": ENC_FORM_AUTO, "n2": ENC_FORM_N2, "shared": ENC_FORM_SHARED, "basen": ENC_FORM_ALWAYS, "always": ENC_FORM_ALWAYS}

# include/zkp_hip.h: the boundary
EXPORTS = {
    # name: (restype, argtypes)
    "zkp_ctx_create": (C.c_int32, [C.c_int32, C.POINTER(C.c_void_p)]),
    "zkp_ctx_destroy": (C.c_int32, [C.c_void_p]),
    "zkp_backend_name": (C.c_char_p, []),
    "zkp_build_limbs_per_lane": (C.c_int32, []),
    "zkp_last_error_string": (C.c_char_p, [C.c_void_p]),
    "zkp_ctx_stream": (C.c_void_p, [C.c_void_p]),
    "zkp_ctx_synchronize": (C.c_int32, [C.c_void_p]),
    "zkp_ctx_release_staging": (C.c_int32, [C.c_void_p]),
    "zkp_timing_reset": (C.c_int32, [C.c_void_p, C.c_int32]),
    "zkp_timing_get": (C.c_int32, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "zkp_modexp_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p,
                                     C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]),
    "zkp_modmul_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_uint64, C.c_void_p, C.c_uint32]),
    "zkp_paillier_enc_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_uint32]),
    "zkp_paillier_enc_check_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "zkp_range_ni_prove_batch": (C.c_int32, [C.c_void_p, C.POINTER(RangeNiProofs), C.POINTER(RangeNiWitness),
                                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "zkp_range_ni_verify_batch": (C.c_int32, [C.c_void_p, C.POINTER(RangeNiProofs), C.c_void_p, C.c_uint32]),
    "zkp_range_generate_encrypted_pairs_batch": (C.c_int32, [C.c_void_p, C.POINTER(RangeNiProofs), C.POINTER(RangeNiWitness), C.c_uint32]),
    "zkp_range_challenge_batch": (C.c_int32, [C.c_void_p, C.POINTER(RangeNiProofs), C.c_void_p, C.c_void_p, C.c_uint32]),
    "zkp_range_generate_proof_batch": (C.c_int32, [C.c_void_p, C.POINTER(RangeNiProofs), C.POINTER(RangeNiWitness), C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_uint32]),
    "zkp_range_verifier_output_batch": (C.c_int32, [C.c_void_p, C.POINTER(RangeNiProofs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "zkp_modinv_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32]),
    "zkp_mul_proof_prove_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64] + [C.c_void_p] * 16 + [C.c_uint32]),
    "zkp_mul_proof_verify_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64] + [C.c_void_p] * 9 + [C.c_uint32]),
    "zkp_correct_message_prove_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64] + [C.c_void_p] * 11 + [C.c_uint32]),
    "zkp_correct_message_verify_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint64] + [C.c_void_p] * 6 + [C.c_uint32]),
    "zkp_decimal_to_limbs_batch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]),
    "zkp_decimal_pitch": (C.c_uint32, [C.c_uint32]),
    "zkp_limbs_to_decimal_batch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]),
    "zkp_json_encrypted_pairs_batch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RangeNiProofs), C.c_void_p, C.c_uint32]),
    "zkp_json_range_proof_batch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(RangeNiProofs), C.c_void_p, C.c_uint32]),
    "zkp_json_range_proof_ni_batch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(RangeNiProofs), C.c_void_p, C.c_uint32]),
    "zkp_json_correct_key_proof_batch": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32]),
    "zkp_correct_key_ni_verify_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]),
    "zkp_dlog_prove_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "zkp_dlog_verify_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "zkp_zero_proof_prove_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p, C.c_uint32]),
    "zkp_zero_proof_verify_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_void_p, C.c_uint32]),
    "zkp_ciphertext_proof_prove_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "zkp_ciphertext_proof_verify_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                      C.c_void_p, C.c_void_p, C.c_uint32]),
    "zkp_verlin_proof_prove_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64] + [C.c_void_p] * 16 + [C.c_uint32]),
    "zkp_verlin_proof_verify_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint64] + [C.c_void_p] * 9 + [C.c_uint32]),
    "zkp_multi_create": (C.c_int32, [C.POINTER(C.c_int32), C.c_uint32, C.POINTER(C.c_void_p)]),
    "zkp_multi_destroy": (C.c_int32, [C.c_void_p]),
    "zkp_multi_size": (C.c_uint32, [C.c_void_p]),
    "zkp_multi_ctx": (C.c_void_p, [C.c_void_p, C.c_uint32]),
    "zkp_multi_last_error_string": (C.c_char_p, [C.c_void_p]),
    "zkp_multi_last_timing": (C.c_int32, [C.c_void_p, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "zkp_multi_last_phases": (C.c_int32, [C.c_void_p, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "zkp_ctx_create_on_stream": (C.c_int32, [C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "zkp_ctx_set_geometry": (C.c_int32, [C.c_void_p, C.c_int32]),
    "zkp_ctx_last_geometry": (C.c_int32, [C.c_void_p]),
    "zkp_ctx_latency_limbs_per_lane": (C.c_int32, [C.c_void_p]),
    "zkp_multi_set_gather": (C.c_int32, [C.c_void_p, C.c_uint32]),
    "zkp_multi_gathered": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "zkp_multi_range_ni_prove_batch": (C.c_int32, [C.c_void_p, C.POINTER(RangeNiProofs), C.POINTER(RangeNiWitness), C.c_void_p, C.c_void_p, C.c_void_p]),
    "zkp_multi_range_ni_verify_batch": (C.c_int32, [C.c_void_p, C.POINTER(RangeNiProofs), C.c_void_p]),
    "zkp_multi_correct_key_ni_verify_batch": (C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
}

Z1_EXTRA_LIMBS = 16
_lib = None


def load():
    """dlopen libzkp_hip.so and attach signatures.  Raises if the library is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(there is no CPU fallback)")
        # torch wheels bundle their own libamdhip64 under the same SONAME as /opt/rocm's: whichever is
        # loaded first serves the whole process, and torch refuses to find GPUs behind a foreign one.
        # So when torch is installed, let it load its runtime first (the C ABI itself does not need torch).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in list(EXPORTS.items()) + list(DIAG_EXPORTS.items()):
            fn = getattr(lib, name)   # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class DecItem(C.Structure):
    """zkp_dec_item (include/zkp_hip.h)"""
    _fields_ = [("text_off", C.c_uint64), ("dst_off", C.c_uint64), ("len", C.c_uint32), ("words", C.c_uint32)]


DEC_OK, DEC_INVALID, DEC_NEGATIVE, DEC_OVERFLOW = 0, 1, 2, 3
BIGINT_DEC, BIGINT_HEX, BIGINT_BYTES = 0, 1, 2
DOC_OK, DOC_INVALID, DOC_HOST_PATH = 0, 2, 3
GATHER_HOST, GATHER_RCCL, GATHER_COPY = 0, 1, 2


def bigint_forms(key_form: int, bare_form: int) -> int:
    """ZKP_BIGINT_FORMS: the text form of ek.n and of the bare BigInts (range, ciphertext), named separately"""
    return (key_form << 4) | bare_form


class ZkpError(RuntimeError):
    pass


def ptr(a):
    """numpy array / torch tensor / None -> void* address."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data
    if hasattr(a, "data_ptr"):   # torch tensor (host or device)
        assert a.is_contiguous()
        return a.data_ptr()
    raise TypeError(type(a))


class MultiContext:
    """zkp_multi: several device contexts behind one caller (host buffers only)."""

    def __init__(self, device_ids):
        self.lib = load()
        ids = (C.c_int32 * len(device_ids))(*device_ids)
        h = C.c_void_p()
        st = self.lib.zkp_multi_create(ids, len(device_ids), C.byref(h))
        if st != ZKP_OK:
            raise ZkpError(f"zkp_multi_create({list(device_ids)}) failed with status {st} (no CPU fallback)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.zkp_multi_destroy(self.h)
            self.h = None

    __del__ = close

    def check(self, st):
        if st != ZKP_OK:
            msg = self.lib.zkp_multi_last_error_string(self.h)
            raise ZkpError(f"status {st}: {msg.decode() if msg else ''}")

    def size(self):
        return self.lib.zkp_multi_size(self.h)

    def last_timing(self):
        """[(ms, lo, hi)] per device context for the most recent batch call"""
        out = []
        for i in range(self.size()):
            ms, lo, hi = C.c_double(), C.c_uint64(), C.c_uint64()
            self.check(self.lib.zkp_multi_last_timing(self.h, i, C.byref(ms), C.byref(lo), C.byref(hi)))
            out.append((ms.value, lo.value, hi.value))
        return out

    def last_phases(self):
        """[(compute_ms, gather_ms)] per device context for the most recent batch call (HIP events on each context's stream)"""
        out = []
        for i in range(self.size()):
            cm, gm = C.c_double(), C.c_double()
            self.check(self.lib.zkp_multi_last_phases(self.h, i, C.byref(cm), C.byref(gm)))
            out.append((cm.value, gm.value))
        return out

    def set_gather(self, mode: int):
        """GATHER_HOST (per-GPU D2H) or GATHER_RCCL (one grouped ncclAllGather per output inside the library: the whole result
        device-resident on every GPU, host arrays filled from one copy)"""
        self.check(self.lib.zkp_multi_set_gather(self.h, mode))

    def gathered(self, device_index: int, which: int):
        """(device pointer, block stride in bytes, total bytes) of gathered output `which` (0 verdict/status, 1 c1, 2 c2)"""
        p, stride, total = C.c_void_p(), C.c_uint64(), C.c_uint64()
        self.check(self.lib.zkp_multi_gathered(self.h, device_index, which, C.byref(p), C.byref(stride), C.byref(total)))
        return p.value, stride.value, total.value

    def range_ni_prove(self, proofs, wit, out_e=None, out_e_len=None, out_status=None):
        self.check(self.lib.zkp_multi_range_ni_prove_batch(self.h, C.byref(proofs), C.byref(wit), ptr(out_e), ptr(out_e_len), ptr(out_status)))

    def range_ni_verify(self, proofs, out_verdict):
        self.check(self.lib.zkp_multi_range_ni_verify_batch(self.h, C.byref(proofs), ptr(out_verdict)))

    def correct_key_ni_verify(self, n_bits, batch, n, sigma, salt: bytes, out_verdict):
        self.check(self.lib.zkp_multi_correct_key_ni_verify_batch(self.h, n_bits, batch, ptr(n), ptr(sigma), salt, len(salt), ptr(out_verdict)))


class Context:
    """One GPU + one stream (zkp_ctx)."""

    def __init__(self, device_id: int = 0, stream=None):
        """stream: a hipStream_t (integer / pointer) the caller owns, e.g. torch.cuda.current_stream().cuda_stream; default: the
        ctx creates its own"""
        self.lib = load()
        h = C.c_void_p()
        if stream is None:
            st = self.lib.zkp_ctx_create(device_id, C.byref(h))
        else:
            st = self.lib.zkp_ctx_create_on_stream(device_id, C.c_void_p(stream), C.byref(h))
        if st != ZKP_OK:
            raise ZkpError(f"zkp_ctx_create(device {device_id}) failed with status {st} "
                           "(no gfx950 GPU / HIP runtime error; there is no CPU fallback)")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.zkp_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def check(self, st):
        if st != ZKP_OK:
            msg = self.lib.zkp_last_error_string(self.h)
            raise ZkpError(f"status {st}: {msg.decode() if msg else ''}")

    def synchronize(self):
        self.check(self.lib.zkp_ctx_synchronize(self.h))

    def set_geometry(self, limbs_per_lane: int):
        """0 = automatic (small calls go to the latency engine), 36 / 9 = always that engine"""
        self.check(self.lib.zkp_ctx_set_geometry(self.h, limbs_per_lane))

    def last_geometry(self) -> int:
        """limbs per lane of the engine the most recent batch call ran on"""
        return self.lib.zkp_ctx_last_geometry(self.h)

    def latency_limbs_per_lane(self) -> int:
        """W of the loaded latency engine (libzkp_hip_lat.so); 0 when it is not there"""
        return self.lib.zkp_ctx_latency_limbs_per_lane(self.h)

    def stream(self):
        return self.lib.zkp_ctx_stream(self.h)

    def timing_reset(self, enable=True):
        self.check(self.lib.zkp_timing_reset(self.h, 1 if enable else 0))

    def timing_get(self):
        ms, launches, modexps = C.c_double(), C.c_uint64(), C.c_uint64()
        self.check(self.lib.zkp_timing_get(self.h, C.byref(ms), C.byref(launches), C.byref(modexps)))
        return ms.value, launches.value, modexps.value

    def diag_table_traffic(self, mode: int, passes: int) -> int:
        """a known amount of window-table traffic (reads: mode 0, writes: mode 1) for PMC calibration -> bytes moved"""
        n = C.c_uint64()
        self.check(self.lib.zkp_diag_table_traffic(self.h, mode, passes, C.byref(n)))
        return n.value

    def set_enc_form(self, form):
        """which Paillier launches run in base-n form (include/zkp_hip_diag.h): "auto" | "n2" | "shared" | "basen" (= always), or the number"""
        self.check(self.lib.zkp_diag_set_enc_form(self.h, ENC_FORMS[form] if isinstance(form, str) else int(form)))

    def set_r2l(self, mode: int):
        """the latency engine's one-Enc-per-wavefront ladder: 0 = never, 1 = the library's rule, 2 = whenever it can run"""
        self.check(self.lib.zkp_diag_set_r2l(self.h, mode))

    def mid_limbs_per_lane(self) -> int:
        """limbs per lane of the mid engine (libzkp_hip_mid.so), 0 when it is not loaded"""
        return self.lib.zkp_diag_mid_limbs_per_lane(self.h)

    def set_split(self, on: bool):
        """calls of 65 ... 96 proofs under one 2048-bit key as two concurrent calls on the mid and the latency engine (on by default)"""
        self.check(self.lib.zkp_diag_set_split(self.h, 1 if on else 0))

    def last_split(self) -> int:
        """proofs of the most recent RangeProofNi call that ran on the latency engine beside the mid engine (0: the call was not split)"""
        return self.lib.zkp_diag_last_split(self.h)

    def set_fuse_hash(self, on: bool):
        """the transcript hash of a one-proof verify as a workgroup of its Enc launch (include/zkp_hip_diag.h); on by default"""
        self.check(self.lib.zkp_diag_set_fuse_hash(self.h, 1 if on else 0))

    def last_fused_hash(self) -> bool:
        """did the most recent verify call carry its transcript hash inside its Enc launch?"""
        return self.lib.zkp_diag_last_fused_hash(self.h) == 1

    def set_key_cache(self, on: bool):
        """keep the constants of the one key of a shared-key call across calls (include/zkp_hip_diag.h); on by default"""
        self.check(self.lib.zkp_diag_set_key_cache(self.h, 1 if on else 0))

    def key_cache_state(self, which: int):
        """(valid, last set-up launch returned early, set-ups computed so far) of constants buffer `which` on the engine that ran last"""
        out = (C.c_uint32 * 3)()
        self.check(self.lib.zkp_diag_key_cache_state(self.h, which, out))
        return bool(out[0]), bool(out[1]), int(out[2])

    def r2l_last(self) -> bool:
        return self.lib.zkp_diag_r2l_last(self.h) == 1

    def set_r2l_lanes(self, lanes: int):
        """lane geometry of that ladder: 0 = the library's rule, 36 = five wavefronts per Enc (36 lanes x 2 limbs each), 12 / 8 = one wavefront"""
        self.check(self.lib.zkp_diag_set_r2l_lanes(self.h, lanes))

    def r2l_lanes_last(self) -> int:
        """geometry of the most recent launch of the ladder (0: the most recent Paillier launch was not one)"""
        return self.lib.zkp_diag_r2l_lanes_last(self.h)

    def last_host_blocks(self) -> int:
        """proof blocks of the most recent RangeProofNi prove / verify call on host arrays (1: not cut)"""
        return self.lib.zkp_diag_last_host_blocks(self.h)

    def enc_form(self) -> int:
        return self.lib.zkp_diag_enc_form(self.h)

    def diag_basen_last(self):
        """(lanes per n-sized integer of the most recent base-n launch or 0, whether its key qualified for the form)"""
        lanes, ok = C.c_int32(), C.c_uint32()
        self.check(self.lib.zkp_diag_basen_last(self.h, C.byref(lanes), C.byref(ok)))
        return lanes.value, bool(ok.value)

    def diag_basen(self, n_bits: int, n_words, op: int, xa=None, xb=None, ya=None, yb=None):
        """one base-n operation on raw 29-bit limbs (include/zkp_hip_diag.h) -> uint32 array of 4 L + 4 words"""
        import numpy as np
        L = 72 * (n_bits // 2048)
        z = np.zeros(L, np.uint32)
        arrs = [np.ascontiguousarray(z if v is None else v, dtype=np.uint32) for v in (xa, xb, ya, yb)]
        assert all(a.shape == (L,) for a in arrs)
        n_words = np.ascontiguousarray(n_words, dtype=np.uint32)
        out = np.zeros(4 * L + 4, np.uint32)
        self.check(self.lib.zkp_diag_basen(self.h, n_bits, n_words.ctypes.data, op, *[a.ctypes.data for a in arrs], out.ctypes.data))
        return out

    # ---- L1 primitives (buffers: numpy arrays = host pointers, torch cuda tensors = device pointers)
    @staticmethod
    def _flags(*arrs):
        dev = [hasattr(a, "is_cuda") and a.is_cuda for a in arrs if a is not None]
        if any(dev) and not all(dev):
            raise ValueError("all buffers of one call must live in the same memory space")
        return ZKP_F_DEVICE_PTRS if dev and dev[0] else 0

    def modexp(self, mod_bits, exp_bits, count, base, exp, exp_stride, mod, mod_stride, out):
        self.check(self.lib.zkp_modexp_batch(self.h, mod_bits, exp_bits, count, ptr(base), ptr(exp), exp_stride,
                                             ptr(mod), mod_stride, ptr(out), self._flags(base, exp, mod, out)))

    def modmul(self, mod_bits, count, a, b, mod, mod_stride, out):
        self.check(self.lib.zkp_modmul_batch(self.h, mod_bits, count, ptr(a), ptr(b), ptr(mod), mod_stride, ptr(out),
                                             self._flags(a, b, mod, out)))

    def paillier_enc(self, n_bits, count, n, n_stride, m, r, out_c):
        self.check(self.lib.zkp_paillier_enc_batch(self.h, n_bits, count, ptr(n), n_stride, ptr(m), ptr(r), ptr(out_c),
                                                   self._flags(n, m, r, out_c)))

    def paillier_enc_check(self, n_bits, count, n, n_stride, m, r, mulc_a, mulc_b, expected, out_ok):
        """out_ok[i] = Enc(m, r) == expected[i]  (or == mulc_a[i]*mulc_b[i] mod n^2 when expected is None)"""
        self.check(self.lib.zkp_paillier_enc_check_batch(self.h, n_bits, count, ptr(n), n_stride, ptr(m), ptr(r), ptr(mulc_a), ptr(mulc_b),
                                                         ptr(expected), ptr(out_ok), self._flags(n, m, r, mulc_a, mulc_b, expected, out_ok)))

    def release_staging(self):
        self.check(self.lib.zkp_ctx_release_staging(self.h))

    def range_ni_prove(self, proofs: RangeNiProofs, wit: RangeNiWitness, out_e, out_e_len, out_status, device: bool):
        self.check(self.lib.zkp_range_ni_prove_batch(self.h, C.byref(proofs), C.byref(wit), ptr(out_e), ptr(out_e_len),
                                                     ptr(out_status), ZKP_F_DEVICE_PTRS if device else 0))

    def range_ni_verify(self, proofs: RangeNiProofs, out_verdict, device: bool):
        self.check(self.lib.zkp_range_ni_verify_batch(self.h, C.byref(proofs), ptr(out_verdict),
                                                      ZKP_F_DEVICE_PTRS if device else 0))

    def correct_key_ni_verify(self, n_bits, batch, n, sigma, salt: bytes, out_verdict):
        sb = (C.c_uint8 * len(salt)).from_buffer_copy(salt) if salt else None
        self.check(self.lib.zkp_correct_key_ni_verify_batch(self.h, n_bits, batch, ptr(n), ptr(sigma),
                                                            C.cast(sb, C.c_void_p) if sb else None, len(salt),
                                                            ptr(out_verdict), self._flags(n, sigma, out_verdict)))

    def dlog_prove(self, n_bits, y_bits, batch, N, g, ni, secret, r, out_x, out_y):
        self.check(self.lib.zkp_dlog_prove_batch(self.h, n_bits, y_bits, batch, ptr(N), ptr(g), ptr(ni), ptr(secret),
                                                 ptr(r), ptr(out_x), ptr(out_y), self._flags(N, g, ni, out_x)))

    def dlog_verify(self, n_bits, y_bits, batch, N, g, ni, x, y, out_verdict):
        self.check(self.lib.zkp_dlog_verify_batch(self.h, n_bits, y_bits, batch, ptr(N), ptr(g), ptr(ni), ptr(x),
                                                  ptr(y), ptr(out_verdict), self._flags(N, g, ni, x, y, out_verdict)))

    def zero_proof_prove(self, n_bits, batch, n, n_stride, c, r, r_prime, out_z, out_a):
        self.check(self.lib.zkp_zero_proof_prove_batch(self.h, n_bits, batch, ptr(n), n_stride, ptr(c), ptr(r), ptr(r_prime), ptr(out_z), ptr(out_a),
                                                       self._flags(n, c, r, r_prime, out_z, out_a)))

    def zero_proof_verify(self, n_bits, batch, n, n_stride, c, z, a, out_verdict):
        self.check(self.lib.zkp_zero_proof_verify_batch(self.h, n_bits, batch, ptr(n), n_stride, ptr(c), ptr(z), ptr(a), ptr(out_verdict),
                                                        self._flags(n, c, z, a, out_verdict)))

    def ciphertext_proof_prove(self, n_bits, batch, n, n_stride, c, x, r, x_prime, r_prime, out_z1, out_z2, out_c_prime):
        self.check(self.lib.zkp_ciphertext_proof_prove_batch(self.h, n_bits, batch, ptr(n), n_stride, ptr(c), ptr(x), ptr(r), ptr(x_prime), ptr(r_prime),
                                                             ptr(out_z1), ptr(out_z2), ptr(out_c_prime), self._flags(n, c, x, r, out_z1)))

    def ciphertext_proof_verify(self, n_bits, batch, n, n_stride, c, z1, z2, c_prime, out_verdict):
        self.check(self.lib.zkp_ciphertext_proof_verify_batch(self.h, n_bits, batch, ptr(n), n_stride, ptr(c), ptr(z1), ptr(z2), ptr(c_prime),
                                                              ptr(out_verdict), self._flags(n, c, z1, z2, c_prime, out_verdict)))

    def verlin_proof_prove(self, n_bits, batch, n, n_stride, c, c_prime, phi_x, witness, nonces, outs):
        """witness = (x, x', x'', r_x); nonces = (a, a', a'', r_a); outs = (phi_a, z, z', z'', r_z)"""
        arrs = [c, c_prime, phi_x, *witness, *nonces, *outs]
        self.check(self.lib.zkp_verlin_proof_prove_batch(self.h, n_bits, batch, ptr(n), n_stride, *[ptr(a) for a in arrs], self._flags(n, *arrs)))

    def verlin_proof_verify(self, n_bits, batch, n, n_stride, c, c_prime, phi_x, phi_a, z, zp, zpp, r_z, out_verdict):
        arrs = [c, c_prime, phi_x, phi_a, z, zp, zpp, r_z, out_verdict]
        self.check(self.lib.zkp_verlin_proof_verify_batch(self.h, n_bits, batch, ptr(n), n_stride, *[ptr(a) for a in arrs], self._flags(n, *arrs)))

    def range_challenge(self, proofs, out_e, out_e_len, device: bool):
        """the Fiat-Shamir challenge of (n, c1, c2) alone: out_e [B][32] left aligned, out_e_len [B]"""
        self.check(self.lib.zkp_range_challenge_batch(self.h, C.byref(proofs), ptr(out_e), ptr(out_e_len), ZKP_F_DEVICE_PTRS if device else 0))

    # ---- interactive RangeProof building blocks (challenge supplied by the caller)
    def range_generate_encrypted_pairs(self, proofs, wit, device: bool):
        self.check(self.lib.zkp_range_generate_encrypted_pairs_batch(self.h, C.byref(proofs), C.byref(wit), ZKP_F_DEVICE_PTRS if device else 0))

    def range_generate_proof(self, proofs, wit, e, e_len, out_status, device: bool):
        self.check(self.lib.zkp_range_generate_proof_batch(self.h, C.byref(proofs), C.byref(wit), ptr(e), ptr(e_len), ptr(out_status),
                                                           ZKP_F_DEVICE_PTRS if device else 0))

    def range_verifier_output(self, proofs, e, e_len, out_verdict, device: bool):
        self.check(self.lib.zkp_range_verifier_output_batch(self.h, C.byref(proofs), ptr(e), ptr(e_len), ptr(out_verdict),
                                                            ZKP_F_DEVICE_PTRS if device else 0))

    # ---- mod_inv, MulProof
    def modinv(self, mod_bits, count, a, mod, mod_stride, out, out_status):
        self.check(self.lib.zkp_modinv_batch(self.h, mod_bits, count, ptr(a), ptr(mod), mod_stride, ptr(out), ptr(out_status),
                                             self._flags(a, mod, out, out_status)))

    def mul_proof_prove(self, n_bits, batch, n, n_stride, e_a, e_b, e_c, a, b, r_a, r_b, r_c, d, r_d, out_f, out_z1, out_z2, out_e_d, out_e_db, out_status):
        arrs = (e_a, e_b, e_c, a, b, r_a, r_b, r_c, d, r_d, out_f, out_z1, out_z2, out_e_d, out_e_db, out_status)
        self.check(self.lib.zkp_mul_proof_prove_batch(self.h, n_bits, batch, ptr(n), n_stride, *[ptr(x) for x in arrs], self._flags(n, *arrs)))

    def mul_proof_verify(self, n_bits, batch, n, n_stride, e_a, e_b, e_c, f, z1, z2, e_d, e_db, out_verdict):
        arrs = (e_a, e_b, e_c, f, z1, z2, e_d, e_db, out_verdict)
        self.check(self.lib.zkp_mul_proof_verify_batch(self.h, n_bits, batch, ptr(n), n_stride, *[ptr(x) for x in arrs], self._flags(n, *arrs)))

    # ---- CorrectMessageProof
    def correct_message_prove(self, n_bits, batch, K, n, n_stride, valid, message, r, e_sim, z_sim, w, out_ct, out_e_vec, out_z_vec, out_a_vec, out_status):
        arrs = (valid, message, r, e_sim if K > 1 else None, z_sim if K > 1 else None, w, out_ct, out_e_vec, out_z_vec, out_a_vec, out_status)
        self.check(self.lib.zkp_correct_message_prove_batch(self.h, n_bits, batch, K, ptr(n), n_stride, *[ptr(x) for x in arrs],
                                                            self._flags(n, *[x for x in arrs if x is not None])))

    def correct_message_verify(self, n_bits, batch, K, n, n_stride, valid, ct, e_vec, z_vec, a_vec, out_verdict):
        arrs = (valid, ct, e_vec, z_vec, a_vec, out_verdict)
        self.check(self.lib.zkp_correct_message_verify_batch(self.h, n_bits, batch, K, ptr(n), n_stride, *[ptr(x) for x in arrs], self._flags(n, *arrs)))

    # ---- wire format: decimal strings <-> limbs, serde_json documents -> SoA batch
    def decimal_to_limbs(self, text: bytes, items, dst, out_status):
        """items: ctypes array of DecItem (host); dst / out_status: numpy (host) arrays"""
        buf = (C.c_char * len(text)).from_buffer_copy(text)
        self.check(self.lib.zkp_decimal_to_limbs_batch(self.h, C.cast(buf, C.c_void_p), len(text), C.cast(items, C.c_void_p), len(items), ptr(dst),
                                                       dst.size, ptr(out_status), 0))

    def decimal_pitch(self, words):
        return self.lib.zkp_decimal_pitch(words)

    def limbs_to_decimal(self, src):
        """src: numpy [count][words] -> list of decimal strings (bytes)"""
        count, words = src.shape
        pitch = self.decimal_pitch(words)
        out = np.zeros((count, pitch), np.uint8); ln = np.zeros(count, np.uint32)
        self.check(self.lib.zkp_limbs_to_decimal_batch(self.h, ptr(src), words, words, count, ptr(out), pitch, ptr(ln), 0))
        return [bytes(out[i, pitch - ln[i]:]) for i in range(count)]

    def _json_docs(self, docs):
        text = b"".join(docs)
        off = np.zeros(len(docs), np.uint64); ln = np.array([len(d) for d in docs], np.uint64)
        off[1:] = np.cumsum(ln)[:-1]
        buf = (C.c_char * max(len(text), 1)).from_buffer_copy(text or b" ")
        return buf, off, ln

    def json_encrypted_pairs(self, docs, proofs, out_status, device: bool):
        buf, off, ln = self._json_docs(docs)
        self.check(self.lib.zkp_json_encrypted_pairs_batch(self.h, C.cast(buf, C.c_void_p), ptr(off), ptr(ln), C.byref(proofs), ptr(out_status),
                                                           ZKP_F_DEVICE_PTRS if device else 0))

    def json_range_proof(self, docs, proofs, out_status, device: bool):
        buf, off, ln = self._json_docs(docs)
        self.check(self.lib.zkp_json_range_proof_batch(self.h, C.cast(buf, C.c_void_p), ptr(off), ptr(ln), C.byref(proofs), ptr(out_status),
                                                       ZKP_F_DEVICE_PTRS if device else 0))

    def json_range_proof_ni(self, docs, forms: int, proofs, out_status):
        """whole RangeProofNi documents -> every field of the (host) batch; forms = bigint_forms(key_form, bare_form): BIGINT_DEC /
        BIGINT_HEX / BIGINT_BYTES for the un-annotated ek.n and, separately, for range / ciphertext.  With a shared key proofs.n is
        the verifier's key (an input)."""
        buf, off, ln = self._json_docs(docs)
        self.check(self.lib.zkp_json_range_proof_ni_batch(self.h, C.cast(buf, C.c_void_p), ptr(off), ptr(ln), forms, C.byref(proofs), ptr(out_status), 0))

    def json_correct_key_proof(self, docs, n_bits, out_sigma, out_status):
        buf, off, ln = self._json_docs(docs)
        self.check(self.lib.zkp_json_correct_key_proof_batch(self.h, C.cast(buf, C.c_void_p), ptr(off), ptr(ln), n_bits, len(docs), ptr(out_sigma),
                                                             ptr(out_status), self._flags(out_sigma, out_status)))
