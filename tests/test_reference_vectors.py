"""Consumer of REFERENCE-PRODUCED vectors: tests/golden/reference_vectors.json, written by tools/reference_vectors
(a cargo project that runs the real zk-paillier crate; it cannot be built in this image — no rustc).

While that file is absent the oracle stays "parity unpinned"  and the file-based tests skip.  So that the
consumer is not dead code, the same checks run on a document of the SAME schema minted from oracle/py_model.py
(`simulated_doc`): that proves the checker exercises every section; it pins nothing about the reference.

What the checks establish once a real file is present:
  to_bytes / compute_digest sections  -> N1 (zero = one 00 byte, no separators)
  enc section                         -> the Enc formula incl. m >= n, r >= n
  range_ni section                    -> Open rows are Enc known answers; the Open/Mask pattern is the FS challenge bit string
                                         (compute_digest + to_bytes + MSB-first order, N2); verdicts; serde wire format
  correct_key_ni section              -> extract_nroot's residue (sigma is deterministic), the MGF, the verdict
  dlog section                        -> verify conventions
  serde section (round 3)             -> the text form of the UN-annotated types (bare curv BigInt, kzen-paillier EncryptionKey,
                                         DLogStatement, CompositeDLogProof): the samples decide which of the three encodings the
                                         whole-document reader (zkp_json_range_proof_ni_batch) is given; the "raw" RangeProofNi
                                         documents then go through it and are verified.
A section of a present file that no check reads is a FAILURE, not a skip."""
import json
import os

import numpy as np
import pytest

import helpers as H
from helpers import pm, L, zkp

PATH = os.path.join(H.ROOT, "tests", "golden", "reference_vectors.json")


def D(s):
    return int(s, 10)


def dec(v):
    return str(v)


def simulated_doc():
    """same schema as tools/reference_vectors/src/main.rs, values from the python model (NOT the reference)"""
    p, q, n = H.test_key(1024)
    d = pm.Drbg(b"simulated-reference")
    doc = {"generator": "simulated from oracle/py_model.py (pins nothing)"}
    doc["to_bytes"] = [{"x": dec(x), "hex": pm.to_bytes(x).hex()} for x in (0, 255, 256, 65536)]
    lists = [[0], [0, 0, 1], [n, 255, n * n], [256, 255, 65536]]
    doc["compute_digest"] = [{"items": [dec(v) for v in l], "digest": dec(pm.compute_digest(l))} for l in lists]
    ms = [0, 1, d.bits(256), n - 1, (1 << 1024) - 1]
    rs = [d.below(n), d.below(n), d.below(n), 1, (1 << 1024) - 1]
    doc["enc"] = {"n": dec(n), "items": [{"m": dec(m), "r": dec(r), "c": dec(pm.enc(n, m, r))} for m, r in zip(ms, rs)]}
    doc["range_ni"] = []
    for honest in (True, False):
        c = H.build_range_case(b"simulated-%d" % honest, [n], 1024, 1, honest=honest, ef=128)[0]
        ct = pm.enc(n, c["x"], c["r"])
        pr = pm.range_ni_prove(n, c["range"], ct, c["x"], c["r"], c["w1"], c["w2"], c["r1"], c["r2"])
        resp = []
        for r_ in pr["responses"]:
            if r_[0] == "open":
                resp.append({"Open": {"w1": dec(r_[1]), "r1": dec(r_[2]), "w2": dec(r_[3]), "r2": dec(r_[4])}})
            else:
                resp.append({"Mask": {"j": r_[1], "masked_x": dec(r_[2]), "masked_r": dec(r_[3])}})
        ok = pm.range_ni_verify(pr, n, ct)
        doc["range_ni"].append({"n": dec(n), "range": dec(c["range"]), "ciphertext": dec(ct), "x": dec(c["x"]), "r": dec(c["r"]), "honest": honest,
                                "encrypted_pairs": {"c1": [dec(v) for v in pr["c1"]], "c2": [dec(v) for v in pr["c2"]]}, "proof": resp,
                                "error_factor": 128, "verify_self": "ok" if ok else "err"})
    sig = pm.correct_key_proof(p, q, b"KZen")
    doc["correct_key_ni"] = [{"p": dec(p), "q": dec(q), "n": dec(n), "salt_hex": b"KZen".hex(), "sigma_vec": [dec(v) for v in sig], "verify": "ok"}]
    g = d.range(2, n - 1); s = d.bits(256)
    ni = pow(pow(g, -1, n), s, n)
    x, y = pm.dlog_prove(n, g, ni, s, d.bits(512))
    doc["dlog"] = [{"N": dec(n), "g": dec(g), "ni": dec(ni), "secret": dec(s), "x": dec(x), "y": dec(y), "verify": "ok"}]
    # the "serde" section: the text forms of the UN-annotated types, with the value next to each sample.  The simulated document is
    # MIXED, the way the two crates are recalled to write them [upstream, unverified]: a bare curv BigInt as a hex string of the
    # big-endian magnitude, kzen-paillier's EncryptionKey through its decimal-string adapter.  The consumer derives the two forms
    # independently (bigint_samples / encryption_key); a real file decides.
    hx = lambda v: v.to_bytes(max(1, (v.bit_length() + 7) // 8), "big").hex()
    doc["serde"] = {"bigint_samples": [{"x": dec(v), "json": hx(v)} for v in (0, 255, 256, n)],
                    "encryption_key": {"n": dec(n), "json": {"n": dec(n)}},
                    "dlog_statement": {"N": dec(n), "g": dec(g), "ni": dec(ni), "json": {"N": hx(n), "g": hx(g), "ni": hx(ni)}},
                    "dlog_proof": {"x": dec(x), "y": dec(y), "json": {"x": hx(x), "y": hx(y)}}}
    doc["signed"] = simulated_signed_section(n, d, hx)
    for c in doc["range_ni"]:
        c["raw"] = {"ek": {"n": dec(n)}, "range": hx(D(c["range"])), "ciphertext": hx(D(c["ciphertext"])), "encrypted_pairs": c["encrypted_pairs"],
                    "proof": c["proof"], "error_factor": c["error_factor"]}
    return doc


SIGNED_MUTATIONS = ["none", "neg_r1_open", "neg_masked_r", "wide_masked_r", "neg_masked_x", "neg_w2_only", "neg_w1_and_w2", "short_responses"]


def simulated_signed_section(n, d, hx):
    """the "signed" section of tools/reference_vectors/src/main.rs from the python model (pins nothing): truncated remainders, mod_pow on
    negative bases, Enc on negative operands, and verify_self of documents with one edited decimal field (tests/signed_cases.py)"""
    import oracle_lib
    import signed_cases as S
    a, m = d.bits(300), d.bits(200) + 1
    rem = [{"a": dec(x), "m": dec(y), "rem": dec(pm.tdiv_r(x, y))} for x, y in ((a, m), (-a, m), (a, -m), (-a, -m), (-m, m))]
    base = d.below(n)
    mod_pow = [{"base": dec(b), "exp": dec(n), "modulus": dec(n * n), "out": dec(pow(b, n, n * n))} for b in (-base, -(base + n * n), base + n * n)]
    r, x = d.below(n), d.bits(256)
    enc = [{"m": dec(mm), "r": dec(rr), "c": dec(pm.enc_signed(n, mm, rr))} for mm, rr in ((-1, 1), (-x, r), (x, -r), (-x, -r), (-n, r), (7, 0), (-7, 0))]
    oracle = oracle_lib.Oracle()
    cases = []
    for name in SIGNED_MUTATIONS:
        proof, _ = S.build(name, oracle)
        assert proof["n"] == n
        raw = json.loads(S.document(proof))
        raw["range"] = hx(proof["range"]); raw["ciphertext"] = hx(proof["ciphertext"])       # the bare BigInts in the file's own (here: hex) form
        cases.append({"name": name, "raw": raw, "verify_self": S.model_verdict(proof)})
    return {"rem": rem, "mod_pow": mod_pow, "enc": enc, "range_ni": cases}


def proof_of_raw(raw, key_enc, enc):
    """a whole-document RangeProofNi (as serde wrote it) -> the python-int proof dict of tests/signed_cases.py"""
    resp = []
    for r in raw["proof"]:
        if "Open" in r:
            o = r["Open"]; resp.append(("open", int(o["w1"]), int(o["r1"]), int(o["w2"]), int(o["r2"])))
        else:
            mk = r["Mask"]; resp.append(("mask", int(mk["j"]), int(mk["masked_x"]), int(mk["masked_r"])))
    return dict(n=decode_bigint(raw["ek"]["n"], key_enc), range=decode_bigint(raw["range"], enc), ciphertext=decode_bigint(raw["ciphertext"], enc),
                error_factor=int(raw["error_factor"]), c1=[int(v) for v in raw["encrypted_pairs"]["c1"]], c2=[int(v) for v in raw["encrypted_pairs"]["c2"]], responses=resp)


def check_signed_section(doc, oracle, forms):
    """SURVEY N4 / N5 pinned by the reference itself: `%`, mod_pow, Enc on negative operands; verify_self on edited documents"""
    sg = doc.get("signed")
    if sg is None:
        return
    assert set(sg) <= {"rem", "mod_pow", "enc", "range_ni"}, "unread part of the signed section"
    for t in sg["rem"]:
        assert pm.tdiv_r(int(t["a"]), int(t["m"])) == int(t["rem"]), "Rust `%` on BigInt is not the truncated remainder the oracles assume"
    for t in sg["mod_pow"]:
        assert pow(int(t["base"]), int(t["exp"]), int(t["modulus"])) == int(t["out"]), "mod_pow on a negative base"
    n = D(doc["enc"]["n"])
    for t in sg["enc"]:
        m, r, c = int(t["m"]), int(t["r"]), int(t["c"])
        assert pm.enc_signed(n, m, r) == c == oracle.enc_decimal(n, m, r), (m < 0, r < 0)
    assert forms is not None, "the signed documents need the serde section to say how the un-annotated fields are written"
    for c in sg["range_ni"]:
        proof = proof_of_raw(c["raw"], *forms)
        want = {"ok": "ok", "err": "err", "panic": "panic"}[c["verify_self"]]
        assert oracle.range_ni_verify_decimal(proof)[0] == want, c["name"]


KNOWN_SECTIONS = {"generator", "to_bytes", "compute_digest", "enc", "range_ni", "correct_key_ni", "dlog", "serde", "signed"}


def _forms(v):
    b = v.to_bytes(max(1, (v.bit_length() + 7) // 8), "big")
    return {zkp.BIGINT_DEC: str(v), zkp.BIGINT_HEX: b.hex(), zkp.BIGINT_BYTES: list(b)}


def _fits(j, v):
    """the encodings under which the JSON value j reads as the integer v"""
    return {e for e, f in _forms(v).items() if (f == j or (isinstance(j, str) and isinstance(f, str) and f.lstrip("0") == j.lower().lstrip("0") and e != zkp.BIGINT_DEC))}


def bigint_encoding_of(doc):
    """which text form the un-annotated bare BigInt takes in THIS file: decided by the samples whose values are known"""
    cands = set(_forms(0))
    for smp in doc["serde"]["bigint_samples"]:
        cands &= _fits(smp["json"], D(smp["x"]))
    assert len(cands) == 1, f"the BigInt samples fit {sorted(cands)} of the known encodings (0 dec, 1 hex, 2 bytes): teach the reader the new form"
    return cands.pop()


def key_encoding_of(doc):
    """the form of EncryptionKey.n, from the encryption_key sample ALONE: kzen-paillier serialises its key through its own adapter,
    which need not be curv's bare-BigInt form (the advisor's round-3 finding: one parameter for both cannot read a mixed document)"""
    ek = doc["serde"]["encryption_key"]
    cands = _fits(ek["json"]["n"], D(ek["n"]))
    assert len(cands) == 1, f"EncryptionKey.n fits {sorted(cands)} of the known encodings"
    return cands.pop()


def decode_bigint(j, enc):
    if enc == zkp.BIGINT_DEC:
        return int(j, 10)
    if enc == zkp.BIGINT_HEX:
        return int(j, 16) if j else 0
    return int.from_bytes(bytes(j), "big")


def check_serde_section(doc):
    """every sample of the un-annotated types decodes to the value printed next to it; no section of the file goes unread"""
    unknown = set(doc) - KNOWN_SECTIONS
    assert not unknown, f"sections {sorted(unknown)} of the reference file are not consumed by any check"
    if "serde" not in doc:
        return None
    enc, key_enc = bigint_encoding_of(doc), key_encoding_of(doc)
    sd = doc["serde"]
    assert set(sd) <= {"bigint_samples", "encryption_key", "dlog_statement", "dlog_proof"}, "unread part of the serde section"
    ek = sd["encryption_key"]
    assert decode_bigint(ek["json"]["n"], key_enc) == D(ek["n"])
    for part, fields in (("dlog_statement", ("N", "g", "ni")), ("dlog_proof", ("x", "y"))):
        for f in fields:
            assert decode_bigint(sd[part]["json"][f], enc) == D(sd[part][f]), (part, f)
    return key_enc, enc


def width_for(n):
    return 1024 if n.bit_length() <= 1024 else 2048 if n.bit_length() <= 2048 else 4096


def responses_of(doc_proof):
    out = []
    for r in doc_proof:
        if "Open" in r:
            o = r["Open"]; out.append(("open", D(o["w1"]), D(o["r1"]), D(o["w2"]), D(o["r2"])))
        else:
            m = r["Mask"]; out.append(("mask", int(m["j"]), D(m["masked_x"]), D(m["masked_r"])))
    return out


def batch_from_case(c):
    """one reference transcript -> a 1-proof SoA batch (host)"""
    n = D(c["n"]); nb = width_for(n); kw = nb // 32
    ef = int(c["error_factor"])
    pb = zkp.RangeBatch(nb, 1, ef, shared_key=True)
    pb.n[0] = L.int_to_limbs(n, kw); pb.range[0] = L.int_to_limbs(D(c["range"]), kw); pb.ciphertext[0] = L.int_to_limbs(D(c["ciphertext"]), 2 * kw)
    pb.c1[0] = L.ints_to_limbs([D(v) for v in c["encrypted_pairs"]["c1"]], 2 * kw)
    pb.c2[0] = L.ints_to_limbs([D(v) for v in c["encrypted_pairs"]["c2"]], 2 * kw)
    for i, r in enumerate(responses_of(c["proof"])):
        if r[0] == "open":
            pb.resp_kind[0, i] = zkp.RESP_OPEN
            for f, v in zip(("resp_w1", "resp_r1", "resp_w2", "resp_r2"), r[1:]):
                getattr(pb, f)[0, i] = L.int_to_limbs(v, kw)
        else:
            pb.resp_kind[0, i] = zkp.RESP_MASK; pb.resp_j[0, i] = r[1]
            pb.resp_w1[0, i] = L.int_to_limbs(r[2], kw); pb.resp_r1[0, i] = L.int_to_limbs(r[3], kw)
    return pb, nb


def check_doc_cpu(doc, oracle):
    """every section against the oracle (C/GMP) and the python model"""
    forms = check_serde_section(doc)
    check_signed_section(doc, oracle, forms)
    for c in doc["range_ni"]:
        if "raw" in c and forms is not None:               # the whole-document form agrees with the fields printed beside it
            raw = c["raw"]
            key_enc, enc = forms
            assert decode_bigint(raw["ek"]["n"], key_enc) == D(c["n"]) and decode_bigint(raw["range"], enc) == D(c["range"])
            assert decode_bigint(raw["ciphertext"], enc) == D(c["ciphertext"])
            assert raw["encrypted_pairs"] == c["encrypted_pairs"] and raw["proof"] == c["proof"] and raw["error_factor"] == c["error_factor"]
    for t in doc["to_bytes"]:
        assert pm.to_bytes(D(t["x"])).hex() == t["hex"]                                   # N1
    for t in doc["compute_digest"]:
        assert pm.compute_digest([D(v) for v in t["items"]]) == D(t["digest"])
    n = D(doc["enc"]["n"]); nb = width_for(n); kw = nb // 32
    items = doc["enc"]["items"]
    got = oracle.paillier_enc(nb, L.ints_to_limbs([n], kw), 0, L.ints_to_limbs([D(i["m"]) for i in items], kw), L.ints_to_limbs([D(i["r"]) for i in items], kw))
    assert L.limbs_to_ints(got) == [D(i["c"]) for i in items]
    assert all(pm.enc(n, D(i["m"]), D(i["r"])) == D(i["c"]) for i in items)
    for c in doc["range_ni"]:
        n = D(c["n"]); c1 = [D(v) for v in c["encrypted_pairs"]["c1"]]; c2 = [D(v) for v in c["encrypted_pairs"]["c2"]]
        resp = responses_of(c["proof"])
        e = pm.fs_challenge(n, c1, c2)
        # the reference prover answered bit i with Open (0) / Mask (1): the pattern IS the challenge bit string (N2)
        assert [pm.challenge_bit(e, i) for i in range(len(resp))] == [int(r[0] == "mask") for r in resp]
        for i, r in enumerate(resp):                                                       # Open rows: Enc known answers
            if r[0] == "open":
                assert pm.enc(n, r[1], r[2]) == c1[i] and pm.enc(n, r[3], r[4]) == c2[i]
            elif c["honest"]:                                                              # Mask rows: masked_x = x + w_j, masked_r = r * r_j % n
                cj = c1[i] if r[1] == 1 else c2[i]
                assert (cj * D(c["ciphertext"])) % (n * n) == pm.enc(n, r[2], r[3])
        assert D(c["ciphertext"]) == pm.enc(n, D(c["x"]), D(c["r"]))
        pb, nb = batch_from_case(c)
        v = np.full(1, 9, np.uint8)
        oracle.range_ni_verify(pb.struct(), v)
        assert (v[0] == zkp.VERDICT_ACCEPT) == (c["verify_self"] == "ok")
        assert c["verify_self"] == ("ok" if c["honest"] else "err")
    for k in doc["correct_key_ni"]:
        p, q, n = D(k["p"]), D(k["q"]), D(k["n"])
        assert p * q == n
        sig = [D(v) for v in k["sigma_vec"]]
        assert sig == pm.correct_key_proof(p, q, bytes.fromhex(k["salt_hex"]))            # extract_nroot's residue + MGF
        nb = width_for(n); kw = nb // 32
        v = oracle.correct_key_ni_verify(nb, L.ints_to_limbs([n], kw), L.ints_to_limbs(sig, kw)[None], bytes.fromhex(k["salt_hex"]))
        assert (v[0] == zkp.VERDICT_ACCEPT) == (k["verify"] == "ok")
    for t in doc["dlog"]:
        N = D(t["N"]); nb = width_for(N); kw = nb // 32
        arr = lambda v, w=kw: L.ints_to_limbs([D(v)], w)
        v = oracle.dlog_verify(nb, 768, arr(t["N"]), arr(t["g"]), arr(t["ni"]), arr(t["x"]), arr(t["y"], 24))
        assert (v[0] == zkp.VERDICT_ACCEPT) == (t["verify"] == "ok")


def check_doc_gpu(doc, ctx):
    """the same sections through the C ABI on the GPU, incl. the serde wire format of the transcripts"""
    forms = check_serde_section(doc)
    if forms is not None:
        # the WHOLE RangeProofNi documents, as the crate wrote them, through zkp_json_range_proof_ni_batch, then verified
        raws = [c for c in doc["range_ni"] if "raw" in c]
        if raws:
            nb = width_for(D(raws[0]["n"]))
            pw = zkp.RangeBatch(nb, len(raws), int(raws[0]["error_factor"]), shared_key=False)
            st = np.full(len(raws), 9, np.uint8)
            ctx.json_range_proof_ni([json.dumps(c["raw"], separators=(",", ":")).encode() for c in raws], zkp.bigint_forms(*forms), pw.struct(), st)
            assert not st.any()
            v = np.full(len(raws), 9, np.uint8)
            ctx.range_ni_verify(pw.struct(), v, device=False)
            assert [int(x) == zkp.VERDICT_ACCEPT for x in v] == [c["verify_self"] == "ok" for c in raws]
            for b, c in enumerate(raws):
                ref, _ = batch_from_case(c)
                for f in ("range", "ciphertext", "c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2"):
                    assert np.array_equal(getattr(pw, f)[b], getattr(ref, f)[0]), f
    if forms is not None and "signed" in doc:
        # the edited documents through the product's host layer (serde_json::range_proof_ni_from_str -> RangeProofNi::verify_batch)
        import subprocess
        import tempfile
        import signed_cases as S
        from test_gpu_host_parity import build_exe
        cases = doc["signed"]["range_ni"]
        with tempfile.NamedTemporaryFile("wb", suffix=".jsonl", delete=False) as f:
            f.write(b"\n".join(S.document(proof_of_raw(c["raw"], *forms)) for c in cases) + b"\n")
        try:
            out = subprocess.run([build_exe()], input=f"range_ni_verify_docs {f.name}\n", capture_output=True, text=True, timeout=900)
        finally:
            os.unlink(f.name)
        assert out.returncode == 0 and out.stdout.split() == [c["verify_self"] for c in cases], (out.stdout, out.stderr)
    n = D(doc["enc"]["n"]); nb = width_for(n); kw = nb // 32
    items = doc["enc"]["items"]
    out = np.zeros((len(items), 2 * kw), np.uint32)
    ctx.paillier_enc(nb, len(items), L.ints_to_limbs([n], kw), 0, L.ints_to_limbs([D(i["m"]) for i in items], kw), L.ints_to_limbs([D(i["r"]) for i in items], kw), out)
    assert L.limbs_to_ints(out) == [D(i["c"]) for i in items]
    for c in doc["range_ni"]:
        pb, nb = batch_from_case(c)
        v = np.full(1, 9, np.uint8)
        ctx.range_ni_verify(pb.struct(), v, device=False)
        assert (v[0] == zkp.VERDICT_ACCEPT) == (c["verify_self"] == "ok")
        # wire format: the reference's own serde_json text of EncryptedPairs / Proof read by the GPU ingestion path
        pj = zkp.RangeBatch(nb, 1, int(c["error_factor"]), shared_key=True)
        st = np.full(1, 9, np.uint8)
        ctx.json_encrypted_pairs([json.dumps(c["encrypted_pairs"], separators=(",", ":")).encode()], pj.struct(), st, device=False)
        assert st[0] == 0 and np.array_equal(pj.c1, pb.c1) and np.array_equal(pj.c2, pb.c2)
        ctx.json_range_proof([json.dumps(c["proof"], separators=(",", ":")).encode()], pj.struct(), st, device=False)
        assert st[0] == 0
        for f in ("resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2"):
            assert np.array_equal(getattr(pj, f), getattr(pb, f)), f
    for k in doc["correct_key_ni"]:
        n = D(k["n"]); nb = width_for(n); kw = nb // 32
        v = np.full(1, 9, np.uint8)
        ctx.correct_key_ni_verify(nb, 1, L.ints_to_limbs([n], kw), L.ints_to_limbs([D(s) for s in k["sigma_vec"]], kw)[None].copy(), bytes.fromhex(k["salt_hex"]), v)
        assert (v[0] == zkp.VERDICT_ACCEPT) == (k["verify"] == "ok")
    for t in doc["dlog"]:
        N = D(t["N"]); nb = width_for(N); kw = nb // 32
        arr = lambda v, w=kw: L.ints_to_limbs([D(v)], w)
        v = np.full(1, 9, np.uint8)
        ctx.dlog_verify(nb, 768, 1, arr(t["N"]), arr(t["g"]), arr(t["ni"]), arr(t["x"]), arr(t["y"], 24), v)
        assert (v[0] == zkp.VERDICT_ACCEPT) == (t["verify"] == "ok")


def test_consumer_on_a_simulated_document(oracle):
    check_doc_cpu(simulated_doc(), oracle)


def test_reference_vectors_against_oracle(oracle):
    if not os.path.exists(PATH):
        pytest.skip("tests/golden/reference_vectors.json absent: PARITY UNPINNED (run tools/reference_vectors with a Rust toolchain)")
    check_doc_cpu(json.load(open(PATH)), oracle)


@pytest.mark.gpu
def test_consumer_on_a_simulated_document_gpu(ctx):
    check_doc_gpu(simulated_doc(), ctx)


@pytest.mark.gpu
def test_reference_vectors_on_gpu(ctx):
    if not os.path.exists(PATH):
        pytest.skip("tests/golden/reference_vectors.json absent: PARITY UNPINNED")
    check_doc_gpu(json.load(open(PATH)), ctx)
