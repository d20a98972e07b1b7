import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Nothing here touches $ZKP_BASEN: which Paillier launches take the base-n form is a property of a ctx (zkp_diag_set_enc_form), and
    # the `ctx` fixture below runs the parity tests under BOTH forms.  Contexts a test creates itself run the library's own routing.


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a box without a GPU skips the gpu-marked tests instead of erroring in their fixtures.  When the gpu
    tests are asked for (`-m gpu`) nothing is skipped: without a gfx950 they fail loudly (there is no CPU fallback to hide behind)."""
    # The two extra kernel families of round 5 — the mid engine (w18-basen) and the latency engine with its r2l ladder off (w9-pair) — run
    # the files that are about the arithmetic; the files about documents, host glue and the other proofs keep the three families they had
    # (the whole matrix would take the driver's GPU run from 10 to 19 minutes).
    # The narrowing is REPORTED (pytest's "deselected" count) and can be switched off: ZKP_TEST_FULL_MATRIX=1 runs every file under all five.
    narrow = ("w18-basen", "w9-pair")
    wide_files = ("test_gpu_l1", "test_golden", "test_gpu_range", "test_gpu_challenge", "test_gpu_correct_key", "test_gpu_dlog", "test_sigma_proofs", "test_verlin_proof", "test_gpu_soak")
    # Round 6 (the suite on a budget, tests/test_gpu_suite_budget.py): the whole-document tests — 5 s of JSON per case, the arithmetic under
    # them covered family by family elsewhere — run under ONE family, the throughput engine in base-n form.
    one_family_tests = ("test_gpu_whole_range_proof_ni_documents", "test_consumer_on_a_simulated_document_gpu")
    if os.environ.get("ZKP_TEST_FULL_MATRIX", "0") in ("", "0"):
        def is_narrowed(it):
            if any(t in it.nodeid for t in one_family_tests) and "w36-basen" not in it.nodeid:
                return True
            return any(f"[{n}" in it.nodeid or f"-{n}]" in it.nodeid or f"[{n}-" in it.nodeid for n in narrow) and not any(w in it.nodeid for w in wide_files)
        dropped = [it for it in items if is_narrowed(it)]
        if dropped:
            config.hook.pytest_deselected(items=dropped)
            items[:] = [it for it in items if not is_narrowed(it)]
    if "gpu" in (config.getoption("-m") or "") or _gpu_present():
        return
    skip = pytest.mark.skip(reason="needs a gfx950 GPU (run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def zkp():
    """the product package (directory name has a hyphen, hence importlib)"""
    return importlib.import_module("zk-paillier_amd")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.Oracle()


# (engine = limbs per lane, Enc form on the throughput engine)
_CTX_PARAMS = [(36, "basen"), (36, "n2"), (18, "basen"), (9, None), (9, "pair")]


@pytest.fixture(scope="session", params=_CTX_PARAMS, ids=["w36-basen", "w36-n2", "w18-basen", "w9", "w9-pair"])
def ctx(zkp, request):
    """one GPU context for the whole -m gpu session; fails loudly without GPU / built library.  Every test that takes it runs
    five times (the mid engine, 18 limbs per lane — libzkp_hip_mid.so, the Paillier calls of 41 ... 64 and 129 ... 192 proofs — with every
    launch in base-n form is the third): pinned to the throughput engine (36 limbs per lane) with every Paillier launch in BASE-n form (csrc/kernels_basen.hpp —
    the kernels that carry the large batches), pinned to it with every launch on the n^2-sized kernels (the product's choice for launches
    that leave SIMDs idle, for keys the form does not take, and for n = 1024), pinned to the latency engine (9; libzkp_hip_lat.so) with its own
    rules (one 2048-bit key and a few proofs: the one-Enc-per-wavefront base-n ladder of csrc/kernels_basen_r2l.hpp), and pinned to it with
    that ladder off (the pair ladder on the n^2-sized product, which still serves per-proof keys and 4096-bit keys) —
    left to itself the library would send these small batches to the latency engine only (tests/test_gpu_geometry.py covers that;
    tests/test_gpu_routing.py covers the library's own choice between the two forms at the sizes where it flips)."""
    geometry, form = request.param
    c = zkp.Context(0)
    c.set_geometry(geometry)               # raises when the engine is not loaded
    if form == "pair":
        c.set_r2l(0)
    elif form is not None:
        c.set_enc_form(form)
    c.test_geometry = geometry
    c.test_form = form
    yield c
    c.close()
