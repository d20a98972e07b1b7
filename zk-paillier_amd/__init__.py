"""zk-paillier_amd — MI355X-native batched Paillier ZK-proof engine (hot path of
ZenGo-X/zk-paillier: RangeProofNi prove/verify, NiCorrectKeyProof verify,
CompositeDLogProof).  Python here is binding + batch plumbing only; all arithmetic runs
in the HIP kernels of csrc/ behind the C ABI of include/zkp_hip.h."""
from . import limbs  # noqa: F401
from .capi import (Context, MultiContext, RangeNiProofs, RangeNiWitness, ZkpError, load, LIB_PATH,  # noqa: F401
                   VERDICT_ACCEPT, VERDICT_REJECT, VERDICT_MALFORMED, RESP_OPEN, RESP_MASK, INV_OK, INV_NONE, INV_DOMAIN, DecItem, DEC_OK, DEC_INVALID, DEC_NEGATIVE, DEC_OVERFLOW, BIGINT_DEC, BIGINT_HEX, BIGINT_BYTES, DOC_OK, DOC_INVALID, DOC_HOST_PATH, bigint_forms, GATHER_HOST, GATHER_RCCL, GATHER_COPY,
                   SECURITY_PARAMETER, CORRECT_KEY_M2, ZKP_F_DEVICE_PTRS, EXPORTS, DIAG_EXPORTS)
from .batch import RangeBatch, make_range_witness  # noqa: F401
