#!/usr/bin/env python3
"""include/zkp_hip.h -> bindings/rust/zkp-hip-sys/src/lib.rs: the `extern "C"` side of the drop-in boundary, every symbol, constant
and struct of the header (the diagnostics of include/zkp_hip_diag.h are deliberately not bound).

There is no rustc in this image, so nothing here is compiled; what keeps the crate honest is tests/test_rust_bindings.py, which parses
the header and the generated Rust INDEPENDENTLY of this script and asserts names, arity, parameter and return types, struct layouts
and constants agree both ways — and that this generator reproduces the committed file byte for byte.

    python tools/gen_rust_sys.py            # rewrite bindings/rust/zkp-hip-sys/src/lib.rs
    python tools/gen_rust_sys.py --check    # exit 1 if the committed file is stale"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "zkp_hip.h")
OUT = os.path.join(ROOT, "bindings", "rust", "zkp-hip-sys", "src", "lib.rs")

SCALARS = {"int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "int64_t": "i64", "double": "f64", "char": "c_char", "void": "c_void"}
RUST_KEYWORDS = {"mod", "type", "ref", "in", "fn", "match", "move", "loop", "use", "impl", "self", "box", "as", "where", "crate", "super", "trait", "struct", "enum"}


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)


def c_type_to_rust(ctype, known_structs):
    """'const uint32_t*' -> '*const u32'"""
    t = ctype.strip()
    stars = t.count("*")
    t = t.replace("*", " ").split()
    const = "const" in t
    base = [w for w in t if w != "const"]
    assert len(base) == 1, ctype
    base = base[0]
    if base in SCALARS:
        r = SCALARS[base]
    elif base in known_structs:
        r = base
    else:
        raise ValueError(f"unknown C type {ctype!r}")
    if stars == 0:
        assert r != "c_void", ctype
        return r
    out = r
    for level in range(stars):
        # only the innermost pointee can be const in this header's style (const T*), outer levels are mutable out-pointers
        out = ("*const " if (const and level == 0) else "*mut ") + out
    return out


def rust_name(name):
    return name + "_" if name in RUST_KEYWORDS else name


def const_type(name):
    if name.startswith(("ZKP_VERDICT_", "ZKP_RESP_", "ZKP_DEC_", "ZKP_INV_", "ZKP_DOC_")):
        return "u8"
    if name in ("ZKP_SECURITY_PARAMETER", "ZKP_CORRECT_KEY_M2", "ZKP_Z1_EXTRA_LIMBS"):
        return "usize"
    return "u32"


def parse_header(text):
    src = strip_comments(text)
    items = []            # in header order: ("const", name, type, value) | ("opaque", name) | ("struct", name, [(field, ctype)]) | ("fn", name, ret, [(param, ctype)]) | ("macro_fn", ...)
    structs = set(re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", src)) | set(re.findall(r"typedef\s+struct\s*\w*\s*\{[^}]*\}\s*(\w+)\s*;", src))
    pos = 0
    pattern = re.compile(
        r"(?P<define>^[ \t]*#define[ \t]+(?P<dname>\w+)(?P<dargs>\([^)]*\))?[ \t]+(?P<dval>[^\n]+)$)"
        r"|(?P<tenum>typedef\s+enum\s*\{(?P<tebody>[^}]*)\}\s*(?P<tename>\w+)\s*;)"
        r"|(?P<enum>enum\s*\{(?P<ebody>[^}]*)\}\s*;)"
        r"|(?P<opaque>typedef\s+struct\s+(?P<oname>\w+)\s+(?P=oname)\s*;)"
        r"|(?P<struct>typedef\s+struct\s*\w*\s*\{(?P<sbody>[^}]*)\}\s*(?P<sname>\w+)\s*;)"
        r"|(?P<fn>^(?P<ret>(?:const\s+)?\w+\s*\**)\s*(?P<fname>zkp_\w+)\s*\((?P<params>[^)]*)\)\s*;)", re.M)
    for m in pattern.finditer(src):
        if m.group("define"):
            name = m.group("dname")
            if name == "ZKP_HIP_H":
                continue
            if m.group("dargs"):
                items.append(("macro_fn", name, m.group("dargs"), m.group("dval").strip()))
            else:
                items.append(("const", name, const_type(name), m.group("dval").strip().rstrip("u")))
        elif m.group("tenum") or m.group("enum"):
            body = m.group("tebody") if m.group("tenum") else m.group("ebody")
            ty = "i32" if m.group("tenum") else None
            for ent in body.split(","):
                ent = ent.strip()
                if not ent:
                    continue
                k, v = [x.strip() for x in ent.split("=")]
                items.append(("const", k, ty or const_type(k), v.rstrip("u")))
        elif m.group("opaque"):
            items.append(("opaque", m.group("oname")))
        elif m.group("struct"):
            fields = []
            for decl in m.group("sbody").split(";"):
                decl = decl.strip()
                if not decl:
                    continue
                ctype, names = re.match(r"(.*?)(\w+(?:\s*,\s*\w+)*)$", decl, re.S).groups()
                # 'uint32_t* c1' : the star belongs to the type
                for nm in names.split(","):
                    fields.append((nm.strip(), ctype.strip()))
            items.append(("struct", m.group("sname"), fields))
        elif m.group("fn"):
            params = []
            ptxt = m.group("params").strip()
            if ptxt and ptxt != "void":
                for prm in ptxt.split(","):
                    prm = " ".join(prm.split())
                    ctype, nm = re.match(r"(.*?)(\w+)$", prm).groups()
                    params.append((nm, ctype.strip()))
            items.append(("fn", m.group("fname"), m.group("ret").strip(), params))
    return items, structs


def generate(text):
    items, structs = parse_header(text)
    out = []
    w = out.append
    w("//! zkp-hip-sys — raw FFI bindings of `libzkp_hip.so`, the MI355X (gfx950) batched Paillier ZK-proof engine.")
    w("//!")
    w("//! GENERATED from `include/zkp_hip.h` by `tools/gen_rust_sys.py`; do not edit.  Every function, struct and constant of the header is")
    w("//! here under its C name (the header's comments cite, per entry point, the line of ZenGo-X/zk-paillier it replaces).  The safe layer")
    w("//! that gives the crate's `zkproofs::{RangeProofNi, NiCorrectKeyProof, CompositeDLogProof}` their GPU paths is `zk-paillier-hip`")
    w("//! (`bindings/rust/zk-paillier-hip/hip.rs`).  `tests/test_rust_bindings.py` checks this file against the header, both ways.")
    w("#![allow(non_camel_case_types, non_snake_case, non_upper_case_globals, clippy::too_many_arguments)]")
    w("")
    w("use std::os::raw::{c_char, c_void};")
    w("")
    consts = [it for it in items if it[0] == "const"]
    w("// ---------------------------------------------------------------- constants (enums and #defines of the header)")
    for _, name, ty, val in consts:
        w(f"pub const {name}: {ty} = {val};")
    for it in items:
        if it[0] == "macro_fn":
            assert it[1] == "ZKP_BIGINT_FORMS", it
            w("/// `ZKP_BIGINT_FORMS(key_form, bare_form)`: the text forms of `ek.n` and of the bare BigInts of a RangeProofNi document")
            w("pub const fn ZKP_BIGINT_FORMS(key_form: u32, bare_form: u32) -> u32 {")
            w("    (key_form << 4) | bare_form")
            w("}")
    w("")
    w("// ---------------------------------------------------------------- opaque handles")
    for it in items:
        if it[0] == "opaque":
            w("#[repr(C)]")
            w(f"pub struct {it[1]} {{")
            w("    _private: [u8; 0],")
            w("}")
    w("")
    w("// ---------------------------------------------------------------- structs (field order and types = the C layout)")
    for it in items:
        if it[0] == "struct":
            w("#[repr(C)]")
            w("#[derive(Clone, Copy, Debug)]")
            w(f"pub struct {it[1]} {{")
            for nm, ctype in it[2]:
                w(f"    pub {rust_name(nm)}: {c_type_to_rust(ctype, structs)},")
            w("}")
    w("")
    w("// ---------------------------------------------------------------- entry points")
    w('#[link(name = "zkp_hip")]')
    w('extern "C" {')
    for it in items:
        if it[0] == "fn":
            _, name, ret, params = it
            ps = ", ".join(f"{rust_name(nm)}: {c_type_to_rust(ct, structs)}" for nm, ct in params)
            rr = "" if ret == "void" else f" -> {c_type_to_rust(ret, structs)}"
            w(f"    pub fn {name}({ps}){rr};")
    w("}")
    return "\n".join(out) + "\n"


DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN extern-block (tools/gen_rust_sys.py rewrites this) -->\n", "<!-- END extern-block -->"


def quoted_in_doc(text):
    """INTEGRATION.md with the extern block of `text` between its markers"""
    doc = open(DOC).read()
    block = text[text.index('#[link(name = "zkp_hip")]'):]
    i, j = doc.index(BEGIN) + len(BEGIN), doc.index(END)
    return doc[:i] + "```rust\n" + block + "```\n" + doc[j:]


def main():
    text = generate(open(HEADER).read())
    if "--check" in sys.argv:
        ok = os.path.exists(OUT) and open(OUT).read() == text and open(DOC).read() == quoted_in_doc(text)
        print("up to date" if ok else f"{OUT} or INTEGRATION.md is stale: run python tools/gen_rust_sys.py")
        sys.exit(0 if ok else 1)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(text)
    doc = quoted_in_doc(text)
    with open(DOC, "w") as f:
        f.write(doc)
    print("wrote", os.path.relpath(OUT, ROOT), f"({text.count('pub fn zkp_')} functions) and the quoted block of INTEGRATION.md")


if __name__ == "__main__":
    main()
