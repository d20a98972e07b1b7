"""Wire format: decimal-string big integers (src/serialize.rs:1-78) and the serde_json documents of
EncryptedPairs / Proof (range_proof.rs:32-81) and NiCorrectKeyProof (correct_key_ni.rs:35-39).
CPU: the GMP oracle (mpz_set_str / mpz_get_str = what BigInt::from_str_radix / to_str_radix call) against Python ints.
GPU: k_dec2bin / k_bin2dec against the oracle byte for byte; JSON ingestion against Python's json + int."""
import ctypes as C
import json
import sys

import numpy as np
import pytest

import helpers as H
from helpers import pm, L, zkp

sys.set_int_max_str_digits(0)   # 512-word values have 4933 decimal digits


def dec_cases(seed, words):
    d = pm.Drbg(seed)
    top = 1 << (32 * words)
    vals = [0, 1, 9, 10, 10**9 - 1, 10**9, 10**9 + 1, 10**18, 2**32 - 1, 2**32, 2**64, top - 1, top // 10, top // 3]
    vals += [d.bits(min(int(b), 32 * words)) for b in (1, 31, 32, 33, 63, 64, 65, 500, 1000, 32 * words - 1, 32 * words)]
    vals += [d.below(top) for _ in range(40)]
    texts = [str(v).encode() for v in vals]
    expect = [zkp.DEC_OK] * len(vals)
    # leading zeros, white space (mpz_set_str ignores it), a sign, empty / malformed strings, overflow
    extra = [(b"000123", 123, zkp.DEC_OK), (b" 12 34\t5\n", 12345, zkp.DEC_OK), (b"-0", 0, zkp.DEC_OK), (b"-17", 0, zkp.DEC_NEGATIVE),
             (b"", 0, zkp.DEC_INVALID), (b"-", 0, zkp.DEC_INVALID), (b"12a3", 0, zkp.DEC_INVALID), (b"+5", 0, zkp.DEC_INVALID), (b"1-2", 0, zkp.DEC_INVALID),
             (b"0x10", 0, zkp.DEC_INVALID), (str(top).encode(), 0, zkp.DEC_OVERFLOW), (str(top * 10**9 + 5).encode(), 0, zkp.DEC_OVERFLOW),
             (b"0" * 3000 + b"7", 7, zkp.DEC_OK)]
    for t, v, e in extra:
        texts.append(t); vals.append(v); expect.append(e)
    return texts, vals, expect


def pack(texts, words, gap=3):
    """-> (text blob, ctypes DecItem array): strings separated by junk bytes, destinations in reverse order"""
    blob = bytearray(); items = (zkp.DecItem * len(texts))()
    for i, t in enumerate(texts):
        blob += b"#" * gap
        items[i].text_off = len(blob); items[i].len = len(t); items[i].words = words
        items[i].dst_off = (len(texts) - 1 - i) * words
        blob += t
    return bytes(blob) + b"##", items


@pytest.mark.parametrize("words", [8, 64, 128])
def test_oracle_decimal_matches_python(oracle, words):
    texts, vals, expect = dec_cases(b"dec-cpu-%d" % words, words)
    blob, items = pack(texts, words)
    dst = np.full((len(texts), words), 0xA5A5A5A5, np.uint32); st = np.full(len(texts), 9, np.uint8)
    oracle.decimal_to_limbs(blob, items, dst, st)
    assert list(st) == expect
    for i, v in enumerate(vals):
        row = dst[len(texts) - 1 - i]
        assert L.limbs_to_int(row) == (v if expect[i] == zkp.DEC_OK else 0), i
    ok = [i for i, e in enumerate(expect) if e == zkp.DEC_OK]
    src = L.ints_to_limbs([vals[i] for i in ok], words)
    out = oracle.limbs_to_decimal(src, 32 * words * 30103 // 100000 + 2)
    assert out == [str(vals[i]).encode() for i in ok]


# ------------------------------------------------------------------ serde_json documents, as serde_json::to_string writes them
def pairs_json(c1, c2, pretty=False):
    doc = {"c1": [str(v) for v in c1], "c2": [str(v) for v in c2]}
    return (json.dumps(doc, indent=2) if pretty else json.dumps(doc, separators=(",", ":"))).encode()


def proof_json(responses, pretty=False):
    rows = []
    for r in responses:
        if r[0] == "open":
            rows.append({"Open": {"w1": str(r[1]), "r1": str(r[2]), "w2": str(r[3]), "r2": str(r[4])}})
        else:
            rows.append({"Mask": {"j": r[1], "masked_x": str(r[2]), "masked_r": str(r[3])}})
    return (json.dumps(rows, indent=2) if pretty else json.dumps(rows, separators=(",", ":"))).encode()


def test_json_fixtures_are_what_the_python_model_reads():
    """the document writers above produce JSON that round-trips through Python's json to the same integers"""
    n = H.test_key(512)[2]
    d = pm.Drbg(b"json-fixture")
    w1, w2, r1, r2 = pm.sample_range_inputs(d, n, 1 << 255, 4)
    c1, c2 = pm.generate_encrypted_pairs(n, w1, w2, r1, r2)
    doc = json.loads(pairs_json(c1, c2, pretty=True))
    assert [int(s) for s in doc["c1"]] == c1 and list(doc) == ["c1", "c2"]
    resp = pm.generate_proof(n, 5, 7, b"\xa0", 1 << 255, w1, w2, r1, r2, 4)
    rows = json.loads(proof_json(resp))
    assert [list(r)[0] for r in rows] == ["Mask", "Open", "Mask", "Open"]


# ================================================================== GPU
@pytest.mark.gpu
@pytest.mark.parametrize("words", [8, 64, 128, 256, 512])
def test_gpu_decimal_matches_oracle(ctx, oracle, words):
    texts, vals, expect = dec_cases(b"dec-gpu-%d" % words, words)
    blob, items = pack(texts, words)
    do = np.full((len(texts), words), 0xA5A5A5A5, np.uint32); so = np.full(len(texts), 9, np.uint8)
    dg = do.copy(); sg = so.copy()
    oracle.decimal_to_limbs(blob, items, do, so)
    ctx.decimal_to_limbs(blob, items, dg, sg)
    assert list(so) == expect and np.array_equal(so, sg) and np.array_equal(do, dg)
    ok = [i for i, e in enumerate(expect) if e == zkp.DEC_OK]
    src = L.ints_to_limbs([vals[i] for i in ok], words)
    assert ctx.limbs_to_decimal(src) == oracle.limbs_to_decimal(src, ctx.decimal_pitch(words)) == [str(vals[i]).encode() for i in ok]


@pytest.mark.gpu
@pytest.mark.parametrize("pretty", [False, True])
def test_gpu_json_ingestion_of_a_proved_batch(ctx, oracle, pretty):
    """prove with the oracle, write every proof as serde_json text, read it back on the GPU: the SoA batch must be identical
    and verify to the same verdicts"""
    n_bits, B, ef = 1024, 6, 128
    kw = n_bits // 32
    n = H.test_key(512)[2]
    cases = H.build_range_case(b"json-gpu", [n], n_bits, B)
    po, wt = H.fill_batch(cases, n_bits, True, oracle)
    oracle.range_ni_prove(po.struct(), wt.struct(), None, None, None)
    pair_docs = [pairs_json([L.limbs_to_int(x) for x in po.c1[b]], [L.limbs_to_int(x) for x in po.c2[b]], pretty) for b in range(B)]
    proof_docs = [proof_json(H.responses_from_batch(po, b), pretty) for b in range(B)]
    pg = zkp.RangeBatch(n_bits, B, ef, shared_key=True)
    pg.n[:] = po.n; pg.range[:] = po.range; pg.ciphertext[:] = po.ciphertext
    for f in ("c1", "c2", "resp_w1", "resp_r1", "resp_w2", "resp_r2"):
        getattr(pg, f)[:] = 0xA5A5A5A5
    pg.resp_kind[:] = 9; pg.resp_j[:] = 9
    s1 = np.full(B, 9, np.uint8); s2 = np.full(B, 9, np.uint8)
    ctx.json_encrypted_pairs(pair_docs, pg.struct(), s1, device=False)
    ctx.json_range_proof(proof_docs, pg.struct(), s2, device=False)
    assert list(s1) == [0] * B and list(s2) == [0] * B
    for f in ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2"):
        assert np.array_equal(getattr(po, f), getattr(pg, f)), f
    v = np.zeros(B, np.uint8)
    ctx.range_ni_verify(pg.struct(), v, device=False)
    assert list(v) == [1] * B


@pytest.mark.gpu
def test_gpu_json_malformed_documents(ctx, oracle):
    n_bits, B, ef = 1024, 8, 4
    kw = n_bits // 32
    n = H.test_key(512)[2]
    d = pm.Drbg(b"json-bad")
    docs, expect = [], []
    good_c = [d.below(n * n) for _ in range(ef)]
    good = pairs_json(good_c, good_c)
    variants = [good, good.replace(b'"c1"', b'"cx"'), good[:-1], good + b"x", pairs_json(good_c[:3], good_c), pairs_json(good_c + [1], good_c),
                good.replace(b'["', b'["-', 1), pairs_json([1 << (64 * kw)] + good_c[1:], good_c)]
    # not JSON of that type: INVALID (2).  Another row count, a negative number, a number wider than the field: valid values of the
    # reference's type that this layout cannot carry: HOST_PATH (3)
    exp = [0, 2, 2, 2, 3, 3, 3, 3]
    pg = zkp.RangeBatch(n_bits, B, ef, shared_key=True)
    st = np.full(B, 9, np.uint8)
    ctx.json_encrypted_pairs(variants, pg.struct(), st, device=False)
    assert list(st) == exp
    assert [L.limbs_to_int(x) for x in pg.c1[0]] == good_c
    assert not pg.c1[1:].any() and not pg.c2[1:].any()   # nothing of a document that is not converted as a whole stays behind
    # Proof documents: wrong variant name, j out of u8 range, missing field, key order
    resp = [("open", 1, 2, 3, 4), ("mask", 2, 5, 6), ("open", 7, 8, 9, 10), ("mask", 1, 11, 12)]
    g = proof_json(resp)
    variants = [g, g.replace(b'"Open"', b'"Opem"', 1), g.replace(b'"j":2', b'"j":256'), g.replace(b'"r2":"4"', b'"r3":"4"'),
                g.replace(b'"w1":"1","r1":"2"', b'"r1":"2","w1":"1"'), proof_json(resp[:3]), g.replace(b'"j":2', b'"j":"2"'), proof_json(resp, pretty=True)]
    st = np.full(B, 9, np.uint8)
    ctx.json_range_proof(variants, pg.struct(), st, device=False)
    assert list(st) == [0, 2, 2, 2, 0, 3, 2, 0]          # (field order inside a Response is free, as it is for serde; 3 rows instead of 4: a valid Proof of another length -> host path)
    assert list(pg.resp_kind[0]) == [0, 1, 0, 1] and list(pg.resp_j[0]) == [0, 2, 0, 1]
    assert [L.limbs_to_int(x) for x in pg.resp_w1[7]] == [1, 5, 7, 11] and [L.limbs_to_int(x) for x in pg.resp_w2[7]] == [3, 0, 9, 0]


@pytest.mark.gpu
def test_gpu_json_correct_key_proof(ctx, oracle):
    n_bits, kw = 1024, 32
    keys = [H.test_key(1024, tag=t) for t in range(3)]
    sig = [pm.correct_key_proof(p_, q_, b"KZen") for p_, q_, _ in keys]
    docs = [json.dumps({"sigma_vec": [str(v) for v in s]}, separators=(",", ":")).encode() for s in sig]
    docs.append(docs[0].replace(b"sigma_vec", b"sigma"))
    docs.append(json.dumps({"sigma_vec": [str(v) for v in sig[0][:10]]}).encode())
    out = np.full((len(docs), 11, kw), 0xA5A5A5A5, np.uint32); st = np.full(len(docs), 9, np.uint8)
    ctx.json_correct_key_proof(docs, n_bits, out, st)
    assert list(st) == [0, 0, 0, 2, 2]
    for b in range(3):
        assert [L.limbs_to_int(x) for x in out[b]] == sig[b]
    n_arr = L.ints_to_limbs([k[2] for k in keys], kw)
    v = np.zeros(3, np.uint8)
    ctx.correct_key_ni_verify(n_bits, 3, n_arr, np.ascontiguousarray(out[:3]), b"KZen", v)
    assert list(v) == [1, 1, 1]


@pytest.mark.gpu
def test_gpu_json_reader_is_as_tolerant_as_serde(ctx):
    """what serde_json + the derived Deserialize impls accept is accepted: fields in any order, unknown fields (any value) skipped,
    escapes inside strings (\\uXXXX digits, escaped field names); what they refuse is refused: duplicate fields, two variant keys."""
    n_bits, ef = 1024, 2
    B = 8
    c1, c2 = [123456789012345678901234567890, 7], [5, 99999999999999999999]
    base = {"c1": [str(v) for v in c1], "c2": [str(v) for v in c2]}
    docs = [
        json.dumps({"c2": base["c2"], "c1": base["c1"]}),                                              # order
        json.dumps({"zz": {"a": [1, 2.5e3, None, True, {"b": "x\\\"y"}]}, "c1": base["c1"], "note": "", "c2": base["c2"]}),   # unknown fields
        '{"c1":["\\u0031\\u00323456789012345678901234567890","7"],"\\u0063\\u0032":["5","99999999999999999999"]}',      # escapes in a number and in a name
        '{"c1":["123456789012345678901234567890","7"],"c2":["5","99999999999999999999"],"c1":["1","2"]}',                 # duplicate field
        '{"c1":["123456789012345678901234567890","7"]}',                                                                    # missing field
        '{"c1":["12345678901234567890123456789\\u0041","7"],"c2":["5","99999999999999999999"]}',                            # \u0041 = "A": not a digit
        json.dumps(base) + " ",                                                                                             # trailing white space
        '{"c1":["1","7"],"c2":["5","9\\"]}',                                                                                # unterminated
    ]
    pg = zkp.RangeBatch(n_bits, B, ef, shared_key=True)
    st = np.full(B, 9, np.uint8)
    ctx.json_encrypted_pairs([d.encode() for d in docs], pg.struct(), st, device=False)
    assert list(st) == [0, 0, 0, 2, 2, 2, 0, 2]
    for b in (0, 1, 2, 6):
        assert [L.limbs_to_int(x) for x in pg.c1[b]] == c1 and [L.limbs_to_int(x) for x in pg.c2[b]] == c2
    rows = [
        '[{"Open":{"r2":"4","w2":"3","r1":"2","w1":"1"}},{"Mask":{"masked_r":"6","j":2,"masked_x":"5","extra":[]}}]',
        '[{"Open":{"w1":"1","r1":"2","w2":"3","r2":"4"},"Mask":{"j":1,"masked_x":"5","masked_r":"6"}},{"Mask":{"j":1,"masked_x":"5","masked_r":"6"}}]',   # two variants
        '[{"Open":{"w1":"1","r1":"2","w2":"3","r2":"4","w1":"1"}},{"Mask":{"j":1,"masked_x":"5","masked_r":"6"}}]',                                       # duplicate
        '[{"Open":{"w1":"1","r1":"2","w2":"3","r2":"4"}},{"Mask":{"j":1.0,"masked_x":"5","masked_r":"6"}}]',                                            # j not an integer
        '[{"Open":{"w1":"\\u0031","r1":"2","w2":"3","r2":"4"}},{"\\u004dask":{"j":0,"masked_x":"5","masked_r":"6"}}]',
    ]
    pr = zkp.RangeBatch(n_bits, len(rows), ef, shared_key=True)
    st = np.full(len(rows), 9, np.uint8)
    ctx.json_range_proof([r.encode() for r in rows], pr.struct(), st, device=False)
    assert list(st) == [0, 2, 2, 2, 0]
    for b in (0, 4):
        assert list(pr.resp_kind[b]) == [0, 1]
        assert [L.limbs_to_int(x) for x in pr.resp_w1[b]] == [1, 5] and [L.limbs_to_int(x) for x in pr.resp_r1[b]] == [2, 6]
        assert [L.limbs_to_int(x) for x in pr.resp_w2[b]] == [3, 0] and [L.limbs_to_int(x) for x in pr.resp_r2[b]] == [4, 0]
    assert list(pr.resp_j[0]) == [0, 2] and list(pr.resp_j[4]) == [0, 0]


def _enc_bigint(v, enc):
    """the three candidate text forms of an un-annotated curv BigInt (include/zkp_hip.h: ZKP_BIGINT_*)"""
    if enc == zkp.BIGINT_DEC:
        return str(v)
    b = v.to_bytes(max(1, (v.bit_length() + 7) // 8), "big")
    return b.hex() if enc == zkp.BIGINT_HEX else list(b)


def range_ni_document(case, pr, enc, ef, pretty=False, extra=False, key_enc=None):
    """serde_json text of a whole RangeProofNi (range_proof_ni.rs:36-44): range / ciphertext in the encoding under test, ek.n in
    `key_enc` (kzen-paillier's EncryptionKey and curv's bare BigInt need not agree), encrypted_pairs / proof in the crate's
    decimal-string format (serialize.rs)"""
    key_enc = enc if key_enc is None else key_enc
    resp = []
    for r in pr["responses"]:
        if r[0] == "open":
            resp.append({"Open": {"w1": str(r[1]), "r1": str(r[2]), "w2": str(r[3]), "r2": str(r[4])}})
        else:
            resp.append({"Mask": {"j": r[1], "masked_x": str(r[2]), "masked_r": str(r[3])}})
    ek = {"n": _enc_bigint(case["n"], key_enc)}
    if extra:
        ek["nn"] = _enc_bigint(case["n"] ** 2, key_enc)           # a fuller EncryptionKey: unknown fields are skipped
    doc = {"ek": ek, "range": _enc_bigint(case["range"], enc), "ciphertext": _enc_bigint(pr["ciphertext"], enc),
           "encrypted_pairs": {"c1": [str(v) for v in pr["c1"]], "c2": [str(v) for v in pr["c2"]]}, "proof": resp, "error_factor": ef}
    return json.dumps(doc, indent=2 if pretty else None, separators=None if pretty else (",", ":")).encode()


@pytest.mark.gpu
@pytest.mark.parametrize("key_enc,enc", [(zkp.BIGINT_DEC, zkp.BIGINT_DEC), (zkp.BIGINT_HEX, zkp.BIGINT_HEX), (zkp.BIGINT_BYTES, zkp.BIGINT_BYTES),
                                         (zkp.BIGINT_DEC, zkp.BIGINT_HEX), (zkp.BIGINT_HEX, zkp.BIGINT_BYTES)], ids=["dec", "hex", "bytes", "key-dec+bare-hex", "key-hex+bare-bytes"])
def test_gpu_whole_range_proof_ni_documents(ctx, oracle, key_enc, enc):
    """whole RangeProofNi documents -> the SoA batch -> verify: the candidate encodings of the un-annotated fields — the key's and the
    bare BigInts' named SEPARATELY (a document may mix them) —, per-proof and shared keys, the documents serde would refuse, and the
    valid ones this layout cannot carry (ZKP_DOC_HOST_PATH)"""
    n_bits, ef, kw = 1024, 128, 32
    forms = zkp.bigint_forms(key_enc, enc)
    keys = [H.test_key(1024, tag=t)[2] for t in range(2)]
    docs, cases = [], []
    for b in range(3):
        c = H.build_range_case(b"whole-doc-%d" % b, [keys[b % 2]], n_bits, 1, honest=(b != 2))[0]
        ct = pm.enc(c["n"], c["x"], c["r"])
        pr = pm.range_ni_prove(c["n"], c["range"], ct, c["x"], c["r"], c["w1"], c["w2"], c["r1"], c["r2"])
        pr["ciphertext"] = ct
        cases.append((c, pr))
        docs.append(range_ni_document(c, pr, enc, ef, pretty=(b == 1), extra=(b == 0), key_enc=key_enc))
    good = docs[0]
    invalid = [good.replace(b'"range"', b'"rnge"'), good.replace(b'"ek"', b'"ek":{"n":"1"},"ek"', 1),
               good[:-1], good.replace(b'"ciphertext":', b'"ciphertext":null,"x":', 1)]
    # valid values of the reference's type that the fixed layout cannot carry: another error_factor, a negative response field
    neg = good.replace(b'"masked_r":"', b'"masked_r":"-', 1)
    host = [good.replace(b'"error_factor":128', b'"error_factor":40'), neg]
    all_docs = docs + invalid + host
    B = len(all_docs)
    pg = zkp.RangeBatch(n_bits, B, ef, shared_key=False)
    st = np.full(B, 9, np.uint8)
    ctx.json_range_proof_ni(all_docs, forms, pg.struct(), st)
    assert list(st) == [0, 0, 0] + [zkp.DOC_INVALID] * len(invalid) + [zkp.DOC_HOST_PATH] * len(host)
    for b, (c, pr) in enumerate(cases):
        assert L.limbs_to_int(pg.n[b]) == c["n"] and L.limbs_to_int(pg.range[b]) == c["range"] and L.limbs_to_int(pg.ciphertext[b]) == pr["ciphertext"]
        assert [L.limbs_to_int(x) for x in pg.c1[b]] == pr["c1"] and [L.limbs_to_int(x) for x in pg.c2[b]] == pr["c2"]
        assert H.responses_from_batch(pg, b) == pr["responses"]
    assert not pg.range[3:].any() and not pg.c1[3:3 + len(invalid)].any() and not pg.n[3:3 + len(invalid)].any()
    v = np.full(3, 9, np.uint8)
    ctx.range_ni_verify(pg.slice(0, 3).struct(), v, device=False)
    assert list(v) == [zkp.VERDICT_ACCEPT, zkp.VERDICT_ACCEPT, zkp.VERDICT_REJECT]
    # one shared key = the VERIFIER's key (an input): the document under the other key is what RangeProofNi::verify's assert_eq!(ek)
    # panics on, and it cannot become the key of the batch even when it comes first
    for order in ([0, 1, 2], [1, 0, 2]):
        ps = zkp.RangeBatch(n_bits, 3, ef, shared_key=True)
        ps.n[0] = L.int_to_limbs(keys[0], kw)
        st = np.full(3, 9, np.uint8)
        ctx.json_range_proof_ni([docs[i] for i in order], forms, ps.struct(), st)
        assert list(st) == [zkp.DOC_INVALID if i == 1 else 0 for i in order] and L.limbs_to_int(ps.n[0]) == keys[0]
    # a value wider than the field, in every encoding: a valid BigInt -> host path
    wide = dict(cases[0][0]); wide["range"] = 1 << 1030
    st = np.full(1, 9, np.uint8)
    pwide = zkp.RangeBatch(n_bits, 1, ef, shared_key=False)          # (kept alive: struct() only borrows the arrays)
    ctx.json_range_proof_ni([range_ni_document(wide, cases[0][1], enc, ef, key_enc=key_enc)], forms, pwide.struct(), st)
    assert list(st) == [zkp.DOC_HOST_PATH] and not pwide.range.any()
    # the form matters: decimal digits read as hex are another number, so a document in the OTHER form must not parse silently into
    # a wrong key — with separate forms the mismatch is either an error or a different (rejected) statement, never the right key
    if key_enc != enc and zkp.BIGINT_BYTES not in (key_enc, enc):
        st = np.full(1, 9, np.uint8)
        pswap = zkp.RangeBatch(n_bits, 1, ef, shared_key=False)
        ctx.json_range_proof_ni([docs[0]], zkp.bigint_forms(enc, key_enc), pswap.struct(), st)
        assert st[0] != 0 or L.limbs_to_int(pswap.n[0]) != cases[0][0]["n"]


@pytest.mark.gpu
def test_gpu_json_readers_into_device_resident_batches(ctx, oracle):
    """ZKP_F_DEVICE_PTRS: the SoA batch and the status bytes live in HBM (documents and offsets stay host text); a document that is
    not converted as a whole — invalid, or valid but outside the layout (a negative number) — leaves only zero rows behind"""
    import torch
    n_bits, ef, kw = 1024, 4, 32
    n = H.test_key(512)[2]
    d = pm.Drbg(b"json-device")
    cs = [d.below(n * n) for _ in range(ef)]
    good = pairs_json(cs, cs[::-1])
    docs = [good, good.replace(b'["', b'["-', 1), good[:-1], good]
    pg = zkp.RangeBatch(n_bits, len(docs), ef, shared_key=True, device="cuda")
    for f in ("c1", "c2"):
        getattr(pg, f).fill_(0x5A5A5A5A)
    st = torch.full((len(docs),), 9, dtype=torch.uint8, device="cuda")
    ctx.json_encrypted_pairs(docs, pg.struct(), st, device=True)
    ctx.synchronize()
    assert st.cpu().tolist() == [zkp.DOC_OK, zkp.DOC_HOST_PATH, zkp.DOC_INVALID, zkp.DOC_OK]
    c1 = pg.c1.cpu().numpy().view(np.uint32); c2 = pg.c2.cpu().numpy().view(np.uint32)
    for b in (0, 3):
        assert [L.limbs_to_int(x) for x in c1[b]] == cs and [L.limbs_to_int(x) for x in c2[b]] == cs[::-1]
    assert not c1[1:3].any() and not c2[1:3].any()
    resp = [("open", 1, 2, 3, 4), ("mask", 2, 5, 6), ("open", 7, 8, 9, 10), ("mask", 1, 11, 12)]
    g = proof_json(resp)
    pdocs = [g, g.replace(b'"masked_r":"6"', b'"masked_r":"-6"'), g.replace(b'"Open"', b'"Opem"', 1), proof_json(resp[:3])]
    for f in ("resp_w1", "resp_r1", "resp_w2", "resp_r2"):
        getattr(pg, f).fill_(0x5A5A5A5A)
    pg.resp_kind.fill_(7); pg.resp_j.fill_(7)
    ctx.json_range_proof(pdocs, pg.struct(), st, device=True)
    ctx.synchronize()
    assert st.cpu().tolist() == [zkp.DOC_OK, zkp.DOC_HOST_PATH, zkp.DOC_INVALID, zkp.DOC_HOST_PATH]
    assert pg.resp_kind.cpu().tolist() == [[0, 1, 0, 1]] + [[0, 0, 0, 0]] * 3 and pg.resp_j.cpu().tolist() == [[0, 2, 0, 1]] + [[0, 0, 0, 0]] * 3
    w1 = pg.resp_w1.cpu().numpy().view(np.uint32)
    assert [L.limbs_to_int(x) for x in w1[0]] == [1, 5, 7, 11] and not w1[1:].any() and not pg.resp_r2.cpu().numpy()[1:].any()
