#!/usr/bin/env python3
"""Writes bindings/rust/zk-paillier-hip.patch: the diff a maintainer of ZenGo-X/zk-paillier 0.4.4 applies (patch -p1) to route the hot
proofs through libzkp_hip.so under cargo feature `hip`.  The edits are stated HERE as (file, anchor, replacement) and applied to a
scratch copy of the reference tree (/root/reference is read, never written); the unified diff of the result is the patch.  The module
the patch wires in is bindings/rust/zk-paillier-hip/hip.rs (copied to src/zkproofs/hip.rs; it is not part of the diff).

    python tools/make_rust_patch.py [--reference /root/reference]

tests/test_rust_bindings.py re-applies the committed patch to a scratch copy of the reference when the tree is present."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "bindings", "rust", "zk-paillier-hip.patch")
FILES = ["Cargo.toml", "src/zkproofs/mod.rs", "src/zkproofs/range_proof.rs", "src/zkproofs/range_proof_ni.rs", "src/zkproofs/correct_key_ni.rs",
         "src/zkproofs/wi_dlog_proof.rs"]

DISPATCH_NOTE = "// feature `hip`: the inner loops run in libzkp_hip.so (MI355X); `None` = not a case for the GPU path, fall through to GMP"

# (file, exact text to find ONCE, text that replaces it)
EDITS = [
    ("Cargo.toml", '[dev-dependencies]\n', '[dependencies.zkp-hip-sys]\npath = "../zk-paillier-amd/bindings/rust/zkp-hip-sys"   # wherever this repository is checked out\noptional = true\n\n[dev-dependencies]\n'),
    ("Cargo.toml", 'default = ["curv-kzen/rust-gmp-kzen"]\n', 'default = ["curv-kzen/rust-gmp-kzen"]\nhip = ["zkp-hip-sys"]\n'),
    ("src/zkproofs/mod.rs", "mod errors;\n", "mod errors;\n#[cfg(feature = \"hip\")]\npub mod hip;\n"),
    # the GPU module builds and reads these values: visible inside `zkproofs`, still private to the outside
    ("src/zkproofs/range_proof.rs", "pub struct Proof(Vec<Response>);", "pub struct Proof(pub(super) Vec<Response>);"),
    ("src/zkproofs/range_proof_ni.rs",
     "    ek: EncryptionKey,\n    range: BigInt,\n    ciphertext: BigInt,\n    encrypted_pairs: EncryptedPairs,\n    proof: Proof,\n    error_factor: usize,\n",
     "    pub(super) ek: EncryptionKey,\n    pub(super) range: BigInt,\n    pub(super) ciphertext: BigInt,\n    pub(super) encrypted_pairs: EncryptedPairs,\n"
     "    pub(super) proof: Proof,\n    pub(super) error_factor: usize,\n"),
    ("src/zkproofs/range_proof_ni.rs", "    ) -> RangeProofNi {\n",
     "    ) -> RangeProofNi {\n        " + DISPATCH_NOTE + "\n        #[cfg(feature = \"hip\")]\n        {\n"
     "            if let Some(proof) = super::hip::range_ni_prove_one(ek, range, ciphertext, secret_x, secret_r) {\n                return proof;\n            }\n        }\n"),
    ("src/zkproofs/range_proof_ni.rs", "        assert_eq!(ciphertext, &self.ciphertext);\n",
     "        assert_eq!(ciphertext, &self.ciphertext);\n        #[cfg(feature = \"hip\")]\n        {\n"
     "            if let Some(verdict) = super::hip::range_ni_verify_one(ek, self) {\n                return verdict;\n            }\n        }\n"),
    ("src/zkproofs/range_proof_ni.rs", "    pub fn verify_self(&self) -> Result<(), IncorrectProof> {\n",
     "    pub fn verify_self(&self) -> Result<(), IncorrectProof> {\n        #[cfg(feature = \"hip\")]\n        {\n"
     "            if let Some(verdict) = super::hip::range_ni_verify_one(&self.ek, self) {\n                return verdict;\n            }\n        }\n"),
    ("src/zkproofs/correct_key_ni.rs", "    pub fn verify(&self, ek: &EncryptionKey, salt_str: &[u8]) -> Result<(), IncorrectProof> {\n",
     "    pub fn verify(&self, ek: &EncryptionKey, salt_str: &[u8]) -> Result<(), IncorrectProof> {\n        " + DISPATCH_NOTE + "\n        #[cfg(feature = \"hip\")]\n        {\n"
     "            if let Some(verdict) = super::hip::correct_key_ni_verify_one(self, ek, salt_str) {\n                return verdict;\n            }\n        }\n"),
    ("src/zkproofs/wi_dlog_proof.rs", "    pub fn prove(statement: &DLogStatement, secret: &BigInt) -> CompositeDLogProof {\n",
     "    pub fn prove(statement: &DLogStatement, secret: &BigInt) -> CompositeDLogProof {\n        " + DISPATCH_NOTE + "\n        #[cfg(feature = \"hip\")]\n        {\n"
     "            if let Some(proof) = super::hip::dlog_prove_one(statement, secret) {\n                return proof;\n            }\n        }\n"),
    ("src/zkproofs/wi_dlog_proof.rs", "    pub fn verify(&self, statement: &DLogStatement) -> Result<(), IncorrectProof> {\n",
     "    pub fn verify(&self, statement: &DLogStatement) -> Result<(), IncorrectProof> {\n        #[cfg(feature = \"hip\")]\n        {\n"
     "            if let Some(verdict) = super::hip::dlog_verify_one(self, statement) {\n                return verdict;\n            }\n        }\n"),
]


def apply_edits(tree):
    for rel, old, new in EDITS:
        path = os.path.join(tree, rel)
        text = open(path).read()
        assert text.count(old) == 1, f"{rel}: anchor {old[:50]!r} occurs {text.count(old)} times in the reference tree"
        with open(path, "w") as f:
            f.write(text.replace(old, new))


def main():
    ref = sys.argv[sys.argv.index("--reference") + 1] if "--reference" in sys.argv else "/root/reference"
    with tempfile.TemporaryDirectory() as tmp:
        for side in ("a", "b"):
            for rel in FILES:
                dst = os.path.join(tmp, side, rel)
                os.makedirs(os.path.dirname(dst), exist_ok=True)
                shutil.copyfile(os.path.join(ref, rel), dst)
        apply_edits(os.path.join(tmp, "b"))
        chunks = []
        for rel in FILES:
            r = subprocess.run(["diff", "-u", "--label", "a/" + rel, "--label", "b/" + rel, os.path.join("a", rel), os.path.join("b", rel)], cwd=tmp, capture_output=True, text=True)
            assert r.returncode == 1, (rel, r.returncode, r.stderr)
            chunks.append(r.stdout)
    header = ("zk-paillier 0.4.4 -> cargo feature `hip`: RangeProofNi::{prove, verify, verify_self}, NiCorrectKeyProof::verify and\n"
              "CompositeDLogProof::{prove, verify} try the GPU path of src/zkproofs/hip.rs first (libzkp_hip.so through the zkp-hip-sys crate)\n"
              "and fall through to their unchanged GMP bodies when it answers None.  Apply with `patch -p1` at the crate root, then copy\n"
              "bindings/rust/zk-paillier-hip/hip.rs to src/zkproofs/hip.rs.  Generated by tools/make_rust_patch.py.\n\n")
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(header + "".join(chunks))
    print("wrote", os.path.relpath(OUT, ROOT))


if __name__ == "__main__":
    main()
