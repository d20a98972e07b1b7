#!/usr/bin/env python3
"""Reads the numeric constants of the hot path out of the reference tree and writes them, with the file:line each one was read from, to
tests/golden/reference_constants.json.

    python tools/pin_reference_constants.py [--reference /root/reference] [--check]

Runs in the BUILD container only (the GPU box has no /root/reference): the JSON it writes is the committed fixture, a table of constants —
no reference source text.  `--check` regenerates in memory and fails when the committed file differs (tests/test_reference_constants.py
runs it that way whenever the reference tree is present).  The reference cannot be compiled here (no rustc), so these constants are the
part of it that CAN be pinned: the oracle (oracle/zkp_oracle.c, oracle/py_model.py), the host layer and the device tables are tested
against this file instead of against values somebody remembered."""
import argparse
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "reference_constants.json")


def find(path, pattern, flags=0):
    """(match, 1-based line of its start) of the first match of `pattern` in the file"""
    text = open(path).read()
    m = re.search(pattern, text, flags)
    if not m:
        raise SystemExit(f"{path}: pattern not found: {pattern}")
    return m, text.count("\n", 0, m.start()) + 1


def collect(ref):
    z = os.path.join(ref, "src", "zkproofs")
    rel = lambda p, line: f"{os.path.relpath(p, ref)}:{line}"
    out = {}

    ck = os.path.join(z, "correct_key_ni.rs")
    m, ln = find(ck, r'const\s+P\s*:\s*&str\s*=\s*"(\d+)"')
    out["correct_key_ni.P"] = {"value": m.group(1), "decimal_digits": len(m.group(1)), "bits": int(m.group(1)).bit_length(), "at": rel(ck, ln)}
    m, ln = find(ck, r"SALT_STRING\s*:\s*&\[u8\]\s*=\s*&\[([0-9,\s]+)\]")
    salt = [int(v) for v in m.group(1).split(",") if v.strip()]
    out["correct_key_ni.SALT_STRING"] = {"value": salt, "ascii": bytes(salt).decode("ascii"), "sha256": hashlib.sha256(bytes(salt)).hexdigest(), "at": rel(ck, ln)}
    for name in ("M2", "DIGEST_SIZE"):
        m, ln = find(ck, rf"const\s+{name}\s*:\s*usize\s*=\s*(\d+)")
        out[f"correct_key_ni.{name}"] = {"value": int(m.group(1)), "at": rel(ck, ln)}

    rn = os.path.join(z, "range_proof_ni.rs")
    m, ln = find(rn, r"const\s+SECURITY_PARAMETER\s*:\s*usize\s*=\s*(\d+)")
    out["range_proof_ni.SECURITY_PARAMETER"] = {"value": int(m.group(1)), "at": rel(rn, ln)}
    m, ln = find(rn, r"const\s+RANGE_BITS\s*:\s*usize\s*=\s*(\d+)")
    out["range_proof_ni.tests.RANGE_BITS"] = {"value": int(m.group(1)), "at": rel(rn, ln)}
    for var in ("p", "q"):
        m, ln = find(rn, rf'let\s+{var}\s*=\s*BigInt::from_str_radix\(\s*"(\d+)"\s*,\s*(\d+)\s*\)', re.S)
        assert int(m.group(2)) == 10
        out[f"range_proof_ni.tests.test_keypair.{var}"] = {"value": m.group(1), "bits": int(m.group(1)).bit_length(), "at": rel(rn, ln)}

    rp = os.path.join(z, "range_proof.rs")
    m, ln = find(rp, r"const\s+STATISTICAL_ERROR_FACTOR\s*:\s*usize\s*=\s*(\d+)")
    out["range_proof.STATISTICAL_ERROR_FACTOR"] = {"value": int(m.group(1)), "at": rel(rp, ln)}

    wd = os.path.join(z, "wi_dlog_proof.rs")
    for name in ("K", "K_PRIME", "SAMPLE_S"):
        m, ln = find(wd, rf"const\s+{name}\s*:\s*usize\s*=\s*(\d+)")
        out[f"wi_dlog_proof.{name}"] = {"value": int(m.group(1)), "at": rel(wd, ln)}

    ck_i = os.path.join(z, "correct_key.rs")
    m, ln = find(ck_i, r"const\s+STATISTICAL_ERROR_FACTOR\s*:\s*usize\s*=\s*(\d+)")
    out["correct_key.STATISTICAL_ERROR_FACTOR"] = {"value": int(m.group(1)), "at": rel(ck_i, ln)}

    cm = os.path.join(z, "correct_message.rs")
    m, ln = find(cm, r"const\s+B\s*:\s*usize\s*=\s*(\d+)")
    out["correct_message.B"] = {"value": int(m.group(1)), "at": rel(cm, ln)}

    bench = os.path.join(ref, "benches", "all.rs")
    if os.path.exists(bench):
        for var in ("p", "q"):
            m, ln = find(bench, rf'let\s+{var}\s*=\s*(?:BigInt::)?(?:from_str_radix|str::parse)?[^"]*"(\d+)"', re.S)
            out[f"benches.test_keypair.{var}"] = {"value": m.group(1), "at": rel(bench, ln)}
    return {"_generated_by": "tools/pin_reference_constants.py (build container; reads /root/reference, which does not travel)",
            "_reference": "ZenGo-X/zk-paillier 0.4.4", "constants": out}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    doc = collect(a.reference)
    text = json.dumps(doc, indent=1, sort_keys=True) + "\n"
    if a.check:
        if not os.path.exists(OUT) or open(OUT).read() != text:
            print("tests/golden/reference_constants.json differs from the reference tree", file=sys.stderr)
            return 1
        return 0
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT, "with", len(doc["constants"]), "constants")
    return 0


if __name__ == "__main__":
    sys.exit(main())
