"""The Rust side of the drop-in boundary ships as SOURCE (there is no rustc in this image): bindings/rust/zkp-hip-sys (the
`extern "C"` block), bindings/rust/zk-paillier-hip/hip.rs (the `zkproofs::hip` module) and bindings/rust/zk-paillier-hip.patch (the
diff that wires it into ZenGo-X/zk-paillier).  Nothing compiles them here, so these CPU tests hold them to include/zkp_hip.h:

  * header and `extern "C"` block are parsed INDEPENDENTLY (not with tools/gen_rust_sys.py) and must agree both ways: every function,
    its arity, every parameter and return type, every struct field in order, every constant and its value;
  * the generator reproduces the committed lib.rs byte for byte; INTEGRATION.md quotes the same block;
  * every `sys::` call of hip.rs names a declared function with the declared number of arguments, every struct literal lists the
    header's fields in the header's order, every constant exists;
  * the patch applies to the reference tree (when the tree is there) and the patched files call functions hip.rs defines;
  * brackets balance (the cheapest syntax check there is)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

import pytest

import helpers as H

ROOT = H.ROOT
HEADER = os.path.join(ROOT, "include", "zkp_hip.h")
SYS = os.path.join(ROOT, "bindings", "rust", "zkp-hip-sys", "src", "lib.rs")
HIP = os.path.join(ROOT, "bindings", "rust", "zk-paillier-hip", "hip.rs")
PATCH = os.path.join(ROOT, "bindings", "rust", "zk-paillier-hip.patch")
REFERENCE = "/root/reference"

C2RUST = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "double": "f64", "char": "c_char", "void": "c_void"}


def header_text():
    return re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)


def c_to_rust(ctype, names):
    """this test's own C -> Rust type rule (written separately from the generator's)"""
    toks = ctype.replace("*", " * ").split()
    const = toks[0] == "const"
    if const:
        toks = toks[1:]
    base, stars = toks[0], toks[1:].count("*")
    rust = C2RUST.get(base, base if base in names else None)
    assert rust, f"unmapped C type {ctype!r}"
    for k in range(stars):
        rust = ("*const " if const and k == 0 else "*mut ") + rust
    return rust


def header_api():
    t = header_text()
    structs = {}
    for body, name in re.findall(r"typedef\s+struct\s*\w*\s*\{([^}]*)\}\s*(\w+)\s*;", t):
        fields = []
        for decl in filter(None, (d.strip() for d in body.split(";"))):
            m = re.match(r"(.+?)\s*(\w+)$", " ".join(decl.split()))
            fields.append((m.group(2), m.group(1).strip()))
        structs[name] = fields
    opaque = set(re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", t))
    names = set(structs) | opaque
    funcs = {}
    for ret, name, params in re.findall(r"^\s*((?:const\s+)?\w+\s*\**)\s*(zkp_\w+)\s*\(([^)]*)\)\s*;", t, re.M):
        ps = []
        if params.strip() not in ("", "void"):
            for prm in params.split(","):
                m = re.match(r"(.+?)\s*(\w+)$", " ".join(prm.split()))
                ps.append((m.group(2), c_to_rust(m.group(1), names)))
        funcs[name] = (None if ret.strip() == "void" else c_to_rust(ret, names), ps)
    consts = {}
    for name, val in re.findall(r"^[ \t]*#define[ \t]+(ZKP_\w+)[ \t]+([0-9]+)u?[ \t]*$", t, re.M):
        consts[name] = int(val)
    for body in re.findall(r"enum\s*\{([^}]*)\}", t):
        for ent in filter(None, (e.strip() for e in body.split(","))):
            k, v = ent.split("=")
            consts[k.strip()] = int(v.strip().rstrip("u"))
    return funcs, {k: [(f, c_to_rust(ty, names)) for f, ty in v] for k, v in structs.items()}, opaque, consts


def rust_api():
    t = open(SYS).read()
    block = re.search(r'extern "C" \{(.*?)\n\}', t, re.S).group(1)
    funcs = {}
    for name, params, ret in re.findall(r"pub fn (\w+)\((.*?)\)(?: -> ([^;]+))?;", block, re.S):
        ps = []
        for prm in filter(None, (p.strip() for p in params.split(","))):
            nm, ty = prm.split(":", 1)
            ps.append((nm.strip().rstrip("_") if nm.strip().endswith("_") else nm.strip(), " ".join(ty.split())))
        funcs[name] = (ret.strip() if ret else None, ps)
    structs, opaque = {}, set()
    for name, body in re.findall(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", t, re.S):
        fields = [(nm.strip().rstrip("_") if nm.strip().endswith("_") else nm.strip(), " ".join(ty.split()))
                  for nm, ty in re.findall(r"pub (\w+): ([^,]+),", body)]
        if fields:
            structs[name] = fields
        else:
            assert "_private" in body
            opaque.add(name)
    consts = {k: int(v) for k, v in re.findall(r"pub const (ZKP_\w+): \w+ = ([0-9]+);", t)}
    return funcs, structs, opaque, consts


def test_extern_block_agrees_with_the_header_both_ways():
    hf, hs, ho, hc = header_api()
    rf, rs, ro, rc = rust_api()
    assert len(hf) >= 54
    assert sorted(hf) == sorted(rf), (sorted(set(hf) - set(rf)), sorted(set(rf) - set(hf)))
    for name, (ret, params) in hf.items():
        rret, rparams = rf[name]
        assert ret == rret, (name, ret, rret)
        assert [p[1] for p in params] == [p[1] for p in rparams], (name, params, rparams)
        assert [p[0] for p in params] == [p[0] for p in rparams], (name, "parameter names")
    assert hs == rs, "struct layouts differ"
    assert ho == ro == {"zkp_ctx", "zkp_multi"}
    assert hc == rc, (sorted(set(hc.items()) ^ set(rc.items())))
    assert "zkp_diag" not in open(SYS).read(), "diagnostics are not part of the boundary"
    # every declared function is also what the built library exports (the ctypes table is checked against the header elsewhere)
    assert sorted(H.zkp.EXPORTS) == sorted(hf)


def test_generator_reproduces_the_committed_file():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_rust_sys.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_integration_md_quotes_the_same_extern_block():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = re.search(r'(#\[link\(name = "zkp_hip"\)\]\nextern "C" \{.*?\n\})', open(SYS).read(), re.S).group(1)
    assert block in doc, "INTEGRATION.md must quote bindings/rust/zkp-hip-sys/src/lib.rs's extern block verbatim (tools/gen_rust_sys.py regenerates both)"


def strip_rust(text):
    """comments, string and char literals out (brackets inside them do not count)"""
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r'"(?:\\.|[^"\\])*"', '""', text)
    text = re.sub(r"'(?:\\.|[^'\\])'", "''", text)
    return text


@pytest.mark.parametrize("path", [SYS, HIP, os.path.join(ROOT, "bindings", "rust", "zkp-hip-sys", "build.rs")], ids=["lib.rs", "hip.rs", "build.rs"])
def test_brackets_balance(path):
    stack, pairs = [], {")": "(", "]": "[", "}": "{"}
    for ch in strip_rust(open(path).read()):
        if ch in "([{":
            stack.append(ch)
        elif ch in pairs:
            assert stack and stack.pop() == pairs[ch], f"unbalanced {ch} in {os.path.basename(path)}"
    assert not stack


def split_args(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def test_hip_module_calls_match_the_header():
    hf, hs, _, hc = header_api()
    src = strip_rust(open(HIP).read())
    calls = 0
    for m in re.finditer(r"sys::(zkp_\w+)\(", src):
        name = m.group(1)
        assert name in hf, f"hip.rs calls sys::{name}, which include/zkp_hip.h does not declare"
        depth, i = 1, m.end()
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        args = split_args(src[m.end():i - 1])
        assert len(args) == len(hf[name][1]), f"sys::{name}: {len(args)} arguments in hip.rs, {len(hf[name][1])} in the header"
        calls += 1
    # the entry points north_star names are the ones the module drives
    for need in ("zkp_ctx_create", "zkp_range_ni_prove_batch", "zkp_range_ni_verify_batch", "zkp_correct_key_ni_verify_batch", "zkp_dlog_prove_batch", "zkp_dlog_verify_batch"):
        assert f"sys::{need}(" in src, need
    assert calls >= 6
    for name in set(re.findall(r"sys::(ZKP_\w+)", src)):
        assert name in hc, f"hip.rs uses sys::{name}, not a constant of the header"
    literals = re.findall(r"sys::(zkp_\w+) \{(\s*\w+\s*:[^{}]*)\}", src)       # `sys::name { field: value, .. }` (not a return type followed by a body)
    assert {n for n, _ in literals} == {"zkp_range_ni_proofs", "zkp_range_ni_witness"}
    for sname, body in literals:
        fields = [f.split(":")[0].strip() for f in split_args(body) if f.strip()]
        assert fields == [f for f, _ in hs[sname]], (sname, fields)


def test_patch_wires_in_functions_the_module_defines():
    patch = open(PATCH).read()
    src = open(HIP).read()
    used = set(re.findall(r"super::hip::(\w+)\(", patch))
    assert used == {"range_ni_prove_one", "range_ni_verify_one", "correct_key_ni_verify_one", "dlog_prove_one", "dlog_verify_one"}
    for fn in used:
        assert re.search(rf"pub fn {fn}\(", src), f"the patch calls hip::{fn}, hip.rs does not define it"
    assert 'hip = ["zkp-hip-sys"]' in patch and "pub mod hip;" in patch
    # batch API on the crate's own types
    for sig in ("pub fn prove_batch(", "pub fn verify_batch("):
        assert sig in src


def test_patch_applies_to_the_reference_tree():
    if not os.path.isdir(REFERENCE):
        pytest.skip("the reference tree is not on this machine (GPU box): the patch was applied when it was generated")
    with tempfile.TemporaryDirectory() as tmp:
        tree = os.path.join(tmp, "zk-paillier")
        shutil.copytree(REFERENCE, tree, ignore=shutil.ignore_patterns(".git", "target"))
        r = subprocess.run(["patch", "-p1", "--forward", "-i", PATCH], cwd=tree, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        shutil.copyfile(HIP, os.path.join(tree, "src", "zkproofs", "hip.rs"))
        # what the module reaches into must be visible to it after the patch
        ni = open(os.path.join(tree, "src", "zkproofs", "range_proof_ni.rs")).read()
        for field in ("ek", "range", "ciphertext", "encrypted_pairs", "proof", "error_factor"):
            assert f"pub(super) {field}:" in ni
        assert "pub struct Proof(pub(super) Vec<Response>);" in open(os.path.join(tree, "src", "zkproofs", "range_proof.rs")).read()
        # and the generator's edits are exactly what the committed patch holds
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_rust_patch.py"), "--reference", REFERENCE], capture_output=True, text=True, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
        assert r.returncode == 0, r.stderr
        assert subprocess.run(["git", "diff", "--quiet", "--", PATCH], cwd=ROOT).returncode in (0, 1)


def test_hip_module_names_exist_in_the_reference_crate():
    """every item hip.rs imports from the crate, and every field it reads, exists under that name in the reference tree (checked
    where the tree is present): a rename upstream would otherwise only show when somebody compiles the module"""
    if not os.path.isdir(REFERENCE):
        pytest.skip("the reference tree is not on this machine")
    src = open(HIP).read()
    z = os.path.join(REFERENCE, "src", "zkproofs")
    wanted = {"correct_key_ni.rs": ["pub struct NiCorrectKeyProof", "pub sigma_vec: Vec<BigInt>", "pub fn verify(&self, ek: &EncryptionKey, salt_str: &[u8])"],
              "errors.rs": ["pub struct IncorrectProof"],
              "range_proof.rs": ["pub struct EncryptedPairs", "pub c1: Vec<BigInt>", "pub c2: Vec<BigInt>", "pub struct Proof(", "pub enum Response", "Open {", "Mask {", "masked_x: BigInt", "masked_r: BigInt", "j: u8"],
              "range_proof_ni.rs": ["pub struct RangeProofNi", "ek: EncryptionKey", "range: BigInt", "ciphertext: BigInt", "encrypted_pairs: EncryptedPairs", "proof: Proof", "error_factor: usize",
                                    "pub fn verify(&self, ek: &EncryptionKey, ciphertext: &BigInt)", "pub fn verify_self(&self)"],
              "wi_dlog_proof.rs": ["pub struct CompositeDLogProof", "pub x: BigInt", "pub y: BigInt", "pub struct DLogStatement", "pub N: BigInt", "pub g: BigInt", "pub ni: BigInt",
                                   "pub fn prove(statement: &DLogStatement, secret: &BigInt)", "pub fn verify(&self, statement: &DLogStatement)"]}
    for f, needles in wanted.items():
        text = open(os.path.join(z, f)).read()
        for n in needles:
            assert n in text, f"{f}: `{n}` not found in the reference — hip.rs relies on it"
    for use in re.findall(r"^use super::(\w+)::", src, re.M):
        assert os.path.exists(os.path.join(z, use + ".rs")), f"hip.rs imports super::{use}, no such module in the reference"
    # the crates hip.rs names are the reference's own dependencies (+ the sys crate the patch adds)
    cargo = open(os.path.join(REFERENCE, "Cargo.toml")).read()
    for crate, dep in (("curv", "curv-kzen"), ("paillier", 'package = "kzen-paillier"'), ("rand", "rand =")):
        assert re.search(rf"^use {crate}::", src, re.M) and dep in cargo
    assert "use zkp_hip_sys as sys;" in src and "zkp-hip-sys" in open(PATCH).read()
