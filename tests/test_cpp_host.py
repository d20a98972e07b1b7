"""The C++ host mirror of the reference API (zk-paillier_amd/host/zkproofs.hpp): it must compile and link
against libzkp_hip.so on any machine (CPU test), and its port of the reference's own unit tests must pass on
the GPU (-m gpu)."""
import os
import subprocess

import pytest

import helpers as H

ROOT = H.ROOT
SRC = os.path.join(ROOT, "tests", "cpp", "test_zkproofs.cpp")
EXE = os.path.join(ROOT, "build", "test_zkproofs")
PKG = os.path.join(ROOT, "zk-paillier_amd")


def build_exe():
    if not os.path.exists(H.zkp.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    deps = [SRC] + [os.path.join(PKG, "host", h) for h in ("zkproofs.hpp", "bigint.hpp", "staging.hpp")] + [H.zkp.LIB_PATH]
    if not os.path.exists(EXE) or any(os.path.getmtime(d) > os.path.getmtime(EXE) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", SRC, "-o", EXE, "-L" + PKG, "-lzkp_hip",
                               "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def test_cpp_mirror_compiles_and_links():
    build_exe()


@pytest.mark.gpu
def test_reference_unit_tests_ported_to_cpp_pass():
    exe = build_exe()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("PASS") == 36 and "FAIL" not in out.stdout


def test_host_bigint_against_gmp():
    """host/bigint.hpp — signed since round 4 (a received proof may hold negative integers): + - * truncated %, floored modulus,
    div_floor, gcd, mod_inv, decimal text, to_bytes against GMP on random mixed-sign operands (tests/cpp/test_bigint.cpp)"""
    src = os.path.join(ROOT, "tests", "cpp", "test_bigint.cpp")
    exe = os.path.join(ROOT, "build", "test_bigint")
    gmp = next((p for p in ("/usr/lib/x86_64-linux-gnu/libgmp.so.10", "/opt/conda/lib/libgmp.so") if os.path.exists(p)), None)
    if gmp is None or not os.path.exists("/opt/conda/include/gmp.h"):
        pytest.skip("no GMP header / library on this machine")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I/opt/conda/include", src, gmp, "-o", exe])
    out = subprocess.run([exe, "8000"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "bigint ok" in out.stdout, out.stdout + out.stderr


def test_staging_pool():
    """host/staging.hpp: blocks of a batch call come back from the pool with their pages mapped, secret ones wiped; bounded; thread-safe
    (tests/cpp/test_staging.cpp; no GPU, no library)"""
    src = os.path.join(ROOT, "tests", "cpp", "test_staging.cpp")
    exe = os.path.join(ROOT, "build", "test_staging")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", src, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "staging ok" in out.stdout, out.stdout + out.stderr
