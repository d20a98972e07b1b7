"""Adversarial RangeProofNi documents with NEGATIVE and over-wide fields: the builder shared by
tests/golden/make_golden.py (which mints tests/golden/signed_cases.json from oracle/py_model.py) and tests/test_signed_values.py
(which replays every case through the C/GMP oracle on the CPU and through host/zkproofs.hpp + the GPU under -m gpu).

A case = an honest proof from a seed (tests/helpers.build_range_case, proved by the C oracle) + one named mutation.  Mutations that
touch c1 / c2 change the Fiat-Shamir challenge (the transcript hash runs over the MAGNITUDES of c1, c2, utils.rs:15-18), so the
"consistent" ones re-answer the new challenge from the witness the way a cheating prover would."""
import hashlib
import json

import helpers as H
from helpers import pm, L

N_BITS = 1024
EF = 128

# name -> what the attacker does
MUTATIONS = [
    "none",                      # the honest proof itself
    "neg_r1_open",               # Open row: r1 - n.  mod_pow(r1, n, nn) is unchanged (result in [0, nn) for a negative base) -> still Ok
    "neg_masked_r",              # Mask row: masked_r - n -> still Ok
    "wide_masked_r",             # Mask row: masked_r + n * 2^1100 -> still Ok
    "neg_masked_x",              # Mask row: masked_x - n: below range/3 -> Err
    "neg_w2_only",               # Open row: w2 - n, c2 untouched: Enc(w2 - n, r2) is NEGATIVE ((m n + 1) % nn keeps the sign), c2 is not -> Err
    "neg_w2_consistent",         # Open row: w2 - n AND c2 - nn, challenge re-answered: Enc matches, `w2 < range/3` holds for a negative w2 -> Ok
    "neg_w1_and_w2",             # Open row: both plaintexts negative: the range flag fails -> Err
    "neg_c1_open",               # Open row: c1 := -c1 (same magnitude: same challenge): compared as signed integers -> Err
    "neg_ciphertext",            # ciphertext - nn: every Mask row's `c_j * ciphertext % nn` turns negative -> Err
    "neg_range",                 # range := -range: range/3 negative, 2*(range/3) < range/3: nothing fits -> Err
    "wide_range",                # range + 2^1500 (wider than the key): thirds far above every w -> Err
    "zero_r_open",               # Open row: r1 := 0 and c1 := 0 (Enc(w, 0) = 0), challenge re-answered -> Ok when the row stays Open
    "short_responses",           # responses truncated to 100 rows: responses[i] panics (range_proof.rs:274)
    "short_c1_mask_j2_rows",     # c1 truncated to 100 entries: rows >= 100 that index c1 panic — unless none of them does (decided by the challenge)
    "other_error_factor",        # error_factor := 64 with 128 rows stored: the first 64 rows are checked, under the challenge over ALL pairs
]


def dec(v):
    return str(v)


def honest_proof(seed: bytes, oracle, n_bits=N_BITS):
    """-> (proof dict of python ints, witness dict)"""
    n = H.test_key(n_bits)[2]
    case = H.build_range_case(seed, [n], n_bits, 1, ef=EF)[0]
    pb, wt = H.fill_batch([case], n_bits, True, oracle)
    oracle.range_ni_prove(pb.struct(), wt.struct(), None, None, None)
    proof = dict(n=n, range=case["range"], ciphertext=L.limbs_to_int(pb.ciphertext[0]), error_factor=EF,
                 c1=[L.limbs_to_int(v) for v in pb.c1[0]], c2=[L.limbs_to_int(v) for v in pb.c2[0]], responses=H.responses_from_batch(pb, 0))
    return proof, case


def answer(proof, wit, overrides):
    """the prover's responses to the challenge of the CURRENT transcript (range_proof.rs:210-252) from the witness; `overrides`
    maps row -> dict(w1=, r1=, w2=, r2=) of values the attacker substituted"""
    n = proof["n"]
    e = pm.fs_challenge_signed(n, proof["c1"], proof["c2"])
    third = proof["range"] // 3
    out = []
    for i in range(EF):
        w = dict(w1=wit["w1"][i], r1=wit["r1"][i], w2=wit["w2"][i], r2=wit["r2"][i])
        w.update(overrides.get(i, {}))
        if not pm.challenge_bit(e, i):
            out.append(("open", w["w1"], w["r1"], w["w2"], w["r2"]))
        elif third < wit["x"] + w["w1"] < 2 * third:
            out.append(("mask", 1, wit["x"] + w["w1"], (wit["r"] * w["r1"]) % n))
        else:
            out.append(("mask", 2, wit["x"] + w["w2"], (wit["r"] * w["r2"]) % n))
    proof["responses"] = out


def first(proof, kind, j=None):
    for i, r in enumerate(proof["responses"]):
        if r[0] == kind and (j is None or r[1] == j):
            return i
    raise AssertionError("no such row")


def mutate(name, proof, wit):
    """applies mutation `name` in place; returns a dict of what was picked (recorded in the golden file)"""
    n, nn = proof["n"], proof["n"] ** 2
    R = proof["responses"]
    if name == "none":
        return {}
    if name == "neg_r1_open":
        i = first(proof, "open"); r = R[i]; R[i] = ("open", r[1], r[2] - n, r[3], r[4]); return {"row": i}
    if name == "neg_masked_r":
        i = first(proof, "mask"); r = R[i]; R[i] = ("mask", r[1], r[2], r[3] - n); return {"row": i}
    if name == "wide_masked_r":
        i = first(proof, "mask"); r = R[i]; R[i] = ("mask", r[1], r[2], r[3] + (n << 1100)); return {"row": i}
    if name == "neg_masked_x":
        i = first(proof, "mask"); r = R[i]; R[i] = ("mask", r[1], r[2] - n, r[3]); return {"row": i}
    if name == "neg_w2_only":
        i = first(proof, "open"); r = R[i]; R[i] = ("open", r[1], r[2], r[3] - n, r[4]); return {"row": i}
    if name == "neg_w1_and_w2":
        i = first(proof, "open"); r = R[i]; R[i] = ("open", r[1] - n, r[2], r[3] - n, r[4]); return {"row": i}
    if name == "neg_c1_open":
        i = first(proof, "open"); proof["c1"][i] = -proof["c1"][i]; return {"row": i}
    if name == "neg_ciphertext":
        proof["ciphertext"] -= nn; return {}
    if name == "neg_range":
        proof["range"] = -proof["range"]; return {}
    if name == "wide_range":
        proof["range"] += 1 << 1500; return {}
    if name == "short_responses":
        del R[100:]; return {}
    if name == "short_c1_mask_j2_rows":
        del proof["c1"][100:]; return {}          # (the challenge now runs over 100 + 128 pairs)
    if name == "other_error_factor":
        proof["error_factor"] = 64; return {}
    if name in ("neg_w2_consistent", "zero_r_open"):
        # substitute row i and re-answer the challenge of the new transcript; keep the first row that the new challenge leaves Open
        base_c1, base_c2 = list(proof["c1"]), list(proof["c2"])
        third = proof["range"] // 3
        for i in range(EF):
            if name == "neg_w2_consistent" and not (wit["w2"][i] < third < wit["w1"][i]):
                continue        # the coin flip left w2 as the LARGE value of this row: `w1 < T && T < w2 < 2T` cannot hold for a negative w2
            proof["c1"], proof["c2"] = list(base_c1), list(base_c2)
            if name == "neg_w2_consistent":
                proof["c2"][i] = base_c2[i] - nn
                ov = {i: dict(w2=wit["w2"][i] - n)}
            else:
                proof["c1"][i] = 0
                ov = {i: dict(r1=0)}
            answer(proof, wit, ov)
            if proof["responses"][i][0] == "open":
                return {"row": i}
        raise AssertionError("no row stays open")
    raise KeyError(name)


def document(proof) -> bytes:
    """serde_json text of the whole RangeProofNi, every integer as a decimal string ('-' for negatives)"""
    resp = []
    for r in proof["responses"]:
        if r[0] == "open":
            resp.append({"Open": {"w1": dec(r[1]), "r1": dec(r[2]), "w2": dec(r[3]), "r2": dec(r[4])}})
        else:
            resp.append({"Mask": {"j": r[1], "masked_x": dec(r[2]), "masked_r": dec(r[3])}})
    doc = {"ek": {"n": dec(proof["n"])}, "range": dec(proof["range"]), "ciphertext": dec(proof["ciphertext"]),
           "encrypted_pairs": {"c1": [dec(v) for v in proof["c1"]], "c2": [dec(v) for v in proof["c2"]]}, "proof": resp,
           "error_factor": proof["error_factor"]}
    return json.dumps(doc, separators=(",", ":")).encode()


def build(name, oracle):
    proof, wit = honest_proof(b"signed-" + name.encode(), oracle)
    picked = mutate(name, proof, wit)
    return proof, picked


def model_verdict(proof):
    """oracle/py_model.range_ni_verify_signed -> "ok" | "err" | "panic" """
    try:
        ok = pm.range_ni_verify_signed(proof["n"], proof["range"], proof["ciphertext"], proof["error_factor"], proof["c1"], proof["c2"], proof["responses"])
    except IndexError:
        return "panic"
    return "ok" if ok else "err"


def sha_doc(proof):
    return hashlib.sha256(document(proof)).hexdigest()
