"""CPU tests of the drop-in boundary: libzkp_hip.so builds/loads and exports every symbol that
include/zkp_hip.h declares; without a GPU the product fails loudly instead of falling back."""
import ctypes as C
import os
import re
import subprocess

import pytest

import helpers as H

ROOT = H.ROOT
zkp = H.zkp


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(zkp.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return zkp.load()


def declared_functions(header="zkp_hip.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zkp_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/zkp_hip.h but not exported by libzkp_hip.so"
        assert n in zkp.EXPORTS, f"{n} has no ctypes signature in capi.EXPORTS"
    assert sorted(zkp.EXPORTS) == names


def test_diagnostics_live_in_their_own_header(lib):
    """the boundary header declares nothing a profiler needs; include/zkp_hip_diag.h is bound separately, symmetrically"""
    diag = declared_functions("zkp_hip_diag.h")
    assert diag and not set(diag) & set(declared_functions()), "a diagnostic is declared in the boundary header"
    assert all(n.startswith("zkp_diag_") for n in diag)
    assert not [n for n in declared_functions() if "diag" in n]
    assert sorted(zkp.DIAG_EXPORTS) == diag
    for n in diag:
        assert hasattr(lib, n)


def test_no_oracle_or_gmp_dependency(lib):
    """the product library must not link the oracle or GMP (there is no CPU fallback path)"""
    out = subprocess.check_output(["ldd", zkp.LIB_PATH], text=True)
    assert "gmp" not in out and "oracle" not in out
    assert lib.zkp_backend_name() == b"hip-gfx950"


def test_struct_layouts_match_header():
    assert C.sizeof(zkp.RangeNiProofs) == 4 + 4 + 8 + 8 + 11 * 8
    assert C.sizeof(zkp.RangeNiWitness) == 6 * 8


def test_ctx_create_fails_loudly_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert lib.zkp_ctx_create(0, C.byref(h)) == zkp.capi.ZKP_EDEVICE and not h.value
    with pytest.raises(zkp.ZkpError):
        zkp.Context(0)


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "zk-paillier_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inc", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), f"{f} mentions the oracle"


def test_c_example_compiles_and_links(lib):
    """examples/multi_gpu_verify.c is plain C against include/zkp_hip.h: the header must be usable from C, the symbols must link"""
    src = os.path.join(ROOT, "examples", "multi_gpu_verify.c")
    exe = os.path.join(ROOT, "build", "multi_gpu_verify")
    pkg = os.path.join(ROOT, "zk-paillier_amd")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), src, "-L" + pkg, "-lzkp_hip",
                           "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    import torch
    if not torch.cuda.is_available():
        out = subprocess.run([exe, "0"], capture_output=True, text=True, timeout=120)
        assert out.returncode == 2 and "no CPU fallback" in out.stderr        # fails loudly without a GPU
