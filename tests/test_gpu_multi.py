"""zkp_multi_*: one caller, several device contexts (include/zkp_hip.h).  The GPU box has ONE GPU, so the device list
repeats device 0: two contexts = two independent streams / scratch sets, which is exactly what two GPUs would be to the
host side (block partition, one thread per context, slabs written into the caller's arrays).  Results must equal the
single-context entry points and the oracle."""
import sys
import numpy as np
import pytest

import helpers as H
from helpers import pm, L, zkp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
def test_multi_range_prove_verify_equals_oracle(oracle, devices):
    n_bits, B = 1024, 5
    n = H.test_key(1024)[2]
    cases = H.build_range_case(b"multi", [n], n_bits, B)
    cases[3] = H.build_range_case(b"multi-bad", [n], n_bits, 1, honest=False)[0]
    pb_o, wt = H.fill_batch(cases, n_bits, True, oracle)
    pb_g = pb_o.to(None)
    oracle.range_ni_prove(pb_o.struct(), wt.struct(), None, None, None)
    m = zkp.MultiContext(devices)
    assert m.size() == len(devices)
    e = np.zeros((B, 32), np.uint8); elen = np.zeros(B, np.uint8); st = np.full(B, 9, np.uint8)
    m.range_ni_prove(pb_g.struct(), wt.struct(), e, elen, st)
    for f in ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2"):
        assert np.array_equal(getattr(pb_o, f), getattr(pb_g, f)), f
    assert not st.any() and (elen > 0).all()
    vg = np.full(B, 9, np.uint8); vo = np.full(B, 9, np.uint8)
    pb_g.resp_r2[1, 5, 3] ^= 4; pb_o.resp_r2[1, 5, 3] ^= 4           # tamper one proof after proving
    m.range_ni_verify(pb_g.struct(), vg)
    oracle.range_ni_verify(pb_o.struct(), vo)
    assert list(vg) == list(vo)
    assert zkp.VERDICT_REJECT in vg and zkp.VERDICT_ACCEPT in vg
    m.close()


def test_multi_per_proof_keys_and_more_contexts_than_proofs(oracle):
    n_bits, B = 1024, 2
    ns = [H.test_key(1024, tag=t)[2] for t in range(B)]
    cases = H.build_range_case(b"multi-keys", ns, n_bits, B, shared=False)
    pb_o, wt = H.fill_batch(cases, n_bits, False, oracle)
    pb_g = pb_o.to(None)
    oracle.range_ni_prove(pb_o.struct(), wt.struct(), None, None, None)
    m = zkp.MultiContext([0, 0, 0])                                   # one context gets an empty block
    m.range_ni_prove(pb_g.struct(), wt.struct())
    assert np.array_equal(pb_o.c1, pb_g.c1) and np.array_equal(pb_o.resp_r1, pb_g.resp_r1)
    v = np.full(B, 9, np.uint8)
    m.range_ni_verify(pb_g.struct(), v)
    assert list(v) == [zkp.VERDICT_ACCEPT] * B
    m.close()


def test_multi_correct_key_verify(oracle):
    keys = [H.test_key(1024, tag=t) for t in range(7)]
    n_arr = L.ints_to_limbs([k[2] for k in keys], 32)
    sig = np.stack([L.ints_to_limbs(pm.correct_key_proof(k[0], k[1], b"KZen"), 32) for k in keys])
    sig[4, 2, 1] ^= 8
    m = zkp.MultiContext([0, 0])
    v = np.full(len(keys), 9, np.uint8)
    m.correct_key_ni_verify(1024, len(keys), n_arr, sig, b"KZen", v)
    assert list(v) == list(oracle.correct_key_ni_verify(1024, n_arr, sig, b"KZen")) == [1, 1, 1, 1, 0, 1, 1]
    m.close()


def test_multi_create_rejects_bad_device_list(zkp):
    with pytest.raises(zkp.ZkpError):
        zkp.MultiContext([0, 99])
    with pytest.raises(zkp.ZkpError):
        zkp.MultiContext([])


def _build_c_example():
    import os
    import subprocess
    src = os.path.join(H.ROOT, "examples", "multi_gpu_verify.c")
    exe = os.path.join(H.ROOT, "build", "multi_gpu_verify")
    pkg = os.path.join(H.ROOT, "zk-paillier_amd")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    if not os.path.exists(exe) or os.path.getmtime(src) > os.path.getmtime(exe) or os.path.getmtime(zkp.LIB_PATH) > os.path.getmtime(exe):
        subprocess.check_call(["gcc", "-O2", "-Wall", "-I" + os.path.join(H.ROOT, "include"), src, "-L" + pkg, "-lzkp_hip",
                               "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_c_example_prove_and_verify_over_three_contexts():
    """examples/multi_gpu_verify.c: a plain-C caller (what the Rust shim of INTEGRATION.md does) proves 64 RangeProofNi, tampers one,
    verifies: 63 accepted, proof 5 rejected"""
    import subprocess
    out = subprocess.run([_build_c_example(), "0", "0", "0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "contexts=3 proofs=64 accepted=63 rejected=1 prove_status_errors=0 verdict[5]=0" in out.stdout
    assert "single-context cross-check: identical" in out.stdout
    assert out.stdout.count("verify context") == 3 and "proofs [0, 22)" in out.stdout          # per-device blocks and timings are printed


def test_c_example_takes_devices_and_batch_from_the_environment():
    import os
    import subprocess
    env = dict(os.environ, ZKP_DEVICES="0,0", ZKP_BATCH="24")
    out = subprocess.run([_build_c_example()], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "contexts=2 proofs=24 accepted=23 rejected=1" in out.stdout and "proofs [12, 24)" in out.stdout


def test_rccl_gather_inside_the_library_equals_the_host_gather(oracle):
    """ZKP_GATHER_RCCL at n_devices = 1 — the only communicator a 1-GPU box can have: the degenerate all-gather still goes through
    real RCCL (ncclCommInitAll, grouped ncclAllGather on the ctx stream), and every byte the caller gets must equal the D2H path's;
    the gathered result is device-resident (read back here through torch)"""
    import torch
    n_bits, B, kw, EF = 1024, 5, 32, 128
    n = H.test_key(1024)[2]
    cases = H.build_range_case(b"multi-rccl", [n], n_bits, B)
    cases[2] = H.build_range_case(b"multi-rccl-bad", [n], n_bits, 1, honest=False)[0]
    pb_h, wt = H.fill_batch(cases, n_bits, True, oracle)
    pb_r = pb_h.to(None)
    m = zkp.MultiContext([0])
    out = {}
    for mode, pb in ((zkp.GATHER_HOST, pb_h), (zkp.GATHER_RCCL, pb_r)):
        m.set_gather(mode)
        e = np.zeros((B, 32), np.uint8); elen = np.zeros(B, np.uint8); st = np.full(B, 9, np.uint8)
        m.range_ni_prove(pb.struct(), wt.struct(), e, elen, st)
        if mode == zkp.GATHER_RCCL:
            for which, host in ((1, pb.c1), (2, pb.c2)):          # c1 / c2 of the whole batch sit in device memory
                dp, stride, total = m.gathered(0, which)
                assert stride == B * EF * 2 * kw * 4 and total == stride
                dev = torch.empty(total, dtype=torch.uint8, device="cuda:0")
                assert torch.cuda.current_stream().synchronize() is None
                import ctypes
                hip = ctypes.CDLL("libamdhip64.so")
                assert hip.hipMemcpy(ctypes.c_void_p(dev.data_ptr()), ctypes.c_void_p(dp), ctypes.c_size_t(total), 3) == 0        # hipMemcpyDeviceToDevice
                assert np.array_equal(dev.cpu().numpy().view(np.uint32).reshape(host.shape), host)
        pb.resp_r1[1, 3, 0] ^= 2
        v = np.full(B, 9, np.uint8)
        m.range_ni_verify(pb.struct(), v)
        out[mode] = (e.copy(), elen.copy(), st.copy(), v.copy())
    for f in ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2"):
        assert np.array_equal(getattr(pb_h, f), getattr(pb_r, f)), f
    for a, b in zip(out[zkp.GATHER_HOST], out[zkp.GATHER_RCCL]):
        assert np.array_equal(a, b)
    assert list(out[zkp.GATHER_RCCL][3]) == [1, 0, 0, 1, 1] and not out[zkp.GATHER_RCCL][2].any()
    # NiCorrectKeyProof verdicts through the same gather
    keys = [H.test_key(1024, tag=t) for t in range(5)]
    n_arr = L.ints_to_limbs([k[2] for k in keys], 32)
    sig = np.stack([L.ints_to_limbs(pm.correct_key_proof(k[0], k[1], b"KZen"), 32) for k in keys])
    sig[1, 0, 0] ^= 1
    v = np.full(5, 9, np.uint8)
    m.correct_key_ni_verify(1024, 5, n_arr, sig, b"KZen", v)
    assert list(v) == [1, 0, 1, 1, 1]
    m.close()


def test_device_resident_gather_over_three_contexts_with_unequal_blocks(oracle):
    """the layout of the device-resident gather at MORE than one block — blocks of 2, 2 and 1 proofs padded to a stride of 2, every
    context holding all three segments, the host arrays reassembled from context 0's copy — on the 1-GPU box: ZKP_GATHER_COPY does the
    exchange of ZKP_GATHER_RCCL with device-to-device copies (RCCL has no communicator for one GPU listed three times)"""
    import ctypes
    import torch
    n_bits, B, kw, EF = 1024, 5, 32, 128
    n = H.test_key(1024)[2]
    cases = H.build_range_case(b"multi-copy", [n], n_bits, B)
    cases[4] = H.build_range_case(b"multi-copy-bad", [n], n_bits, 1, honest=False)[0]
    pb_h, wt = H.fill_batch(cases, n_bits, True, oracle)
    pb_c = pb_h.to(None)
    m = zkp.MultiContext([0, 0, 0])
    res = {}
    for mode, pb in ((zkp.GATHER_HOST, pb_h), (zkp.GATHER_COPY, pb_c)):
        m.set_gather(mode)
        e = np.zeros((B, 32), np.uint8); elen = np.zeros(B, np.uint8); st = np.full(B, 9, np.uint8)
        m.range_ni_prove(pb.struct(), wt.struct(), e, elen, st)
        if mode == zkp.GATHER_COPY:
            assert [(lo, hi) for _, lo, hi in m.last_timing()] == [(0, 2), (2, 4), (4, 5)]
            hip = ctypes.CDLL("libamdhip64.so")
            row = EF * 2 * kw                                    # words of c1 per proof
            for ctx_i in range(3):
                dp, stride, total = m.gathered(ctx_i, 1)
                assert stride == 2 * row * 4 and total == 3 * stride
                dev = torch.empty(total, dtype=torch.uint8, device="cuda:0")
                assert hip.hipMemcpy(ctypes.c_void_p(dev.data_ptr()), ctypes.c_void_p(dp), ctypes.c_size_t(total), 3) == 0
                g = dev.cpu().numpy().view(np.uint32).reshape(3, 2, EF, 2 * kw)          # [block][slot in block][row][limbs]
                assert np.array_equal(g[0], pb.c1[0:2]) and np.array_equal(g[1], pb.c1[2:4]) and np.array_equal(g[2, 0], pb.c1[4])
        pb.resp_r2[3, 7, 1] ^= 16
        v = np.full(B, 9, np.uint8)
        m.range_ni_verify(pb.struct(), v)
        res[mode] = (e.copy(), elen.copy(), st.copy(), v.copy())
    for f in ("c1", "c2", "resp_kind", "resp_j", "resp_w1", "resp_r1", "resp_w2", "resp_r2"):
        assert np.array_equal(getattr(pb_h, f), getattr(pb_c, f)), f
    for a, b in zip(res[zkp.GATHER_HOST], res[zkp.GATHER_COPY]):
        assert np.array_equal(a, b)
    assert list(res[zkp.GATHER_COPY][3]) == [1, 1, 1, 0, 0]
    dp, stride, total = m.gathered(2, 0)                         # the verdict bytes, as context 2 holds them: [1 1 | 1 0 | 0 pad]
    assert (stride, total) == (2, 6)
    keys = [H.test_key(1024, tag=t) for t in range(4)]
    n_arr = L.ints_to_limbs([k[2] for k in keys], 32)
    sig = np.stack([L.ints_to_limbs(pm.correct_key_proof(k[0], k[1], b"KZen"), 32) for k in keys])
    sig[3, 1, 0] ^= 2
    v = np.full(4, 9, np.uint8)
    m.correct_key_ni_verify(1024, 4, n_arr, sig, b"KZen", v)
    assert list(v) == [1, 1, 1, 0]
    m.close()


def test_rccl_gather_needs_distinct_devices():
    """a device listed twice has no RCCL communicator: ZKP_EDEVICE with a text, and the context set keeps working on the host gather"""
    m = zkp.MultiContext([0, 0])
    with pytest.raises(zkp.ZkpError, match="listed twice"):
        m.set_gather(zkp.GATHER_RCCL)
    m.close()


def test_c_example_with_the_rccl_gather():
    import os
    import subprocess
    env = dict(os.environ, ZKP_GATHER="rccl", ZKP_DEVICES="0", ZKP_BATCH="24")
    out = subprocess.run([_build_c_example()], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "contexts=1 proofs=24 accepted=23 rejected=1" in out.stdout and "gather=rccl context 0" in out.stdout and "single-context cross-check: identical" in out.stdout


def test_library_loads_rccl_on_demand_only():
    """RCCL is not a load-time dependency (a single-GPU user needs none installed); the first zkp_multi_set_gather(RCCL) of the process
    brings it in — the other RCCL tests of this file show that it then works"""
    import subprocess
    assert "librccl" not in subprocess.check_output(["ldd", zkp.LIB_PATH], text=True)
    # (plain ctypes in a fresh interpreter: importing the package would import torch, which maps its own bundled librccl)
    code = ("import ctypes as C;"
            "maps = lambda: 'librccl' in open('/proc/self/maps').read();"
            "lib = C.CDLL(%r); ids = (C.c_int32 * 1)(0); h = C.c_void_p();"
            "assert lib.zkp_multi_create(ids, 1, C.byref(h)) == 0; before = maps();"
            "assert lib.zkp_multi_set_gather(h, 1) == 0; after = maps(); lib.zkp_multi_destroy(h); print(before, after)" % zkp.LIB_PATH)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "False True" in out.stdout.splitlines(), out.stdout + out.stderr      # (RCCL prints its version banner after it)


def test_multi_last_timing_reports_blocks(oracle):
    keys = [H.test_key(1024, tag=t) for t in range(5)]
    n_arr = L.ints_to_limbs([k[2] for k in keys], 32)
    sig = np.stack([L.ints_to_limbs(pm.correct_key_proof(k[0], k[1], b"KZen"), 32) for k in keys])
    m = zkp.MultiContext([0, 0])
    v = np.full(len(keys), 9, np.uint8)
    m.correct_key_ni_verify(1024, len(keys), n_arr, sig, b"KZen", v)
    t = m.last_timing()
    assert [(lo, hi) for _, lo, hi in t] == [(0, 3), (3, 5)] and all(ms > 0 for ms, _, _ in t)
    m.close()


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_share_the_one_gpu(scaling):
    """`bench.py --gpus 2` end to end on the 1-GPU box: both ranks on cuda:0, collectives over gloo (ZKP_BENCH_SHARED_GPU=1) — the
    whole N > 1 code path (rank blocks of unequal size, every sharded leg, the gathers, the max-over-ranks timing, the JSON line)
    with real kernels; only RCCL itself is not exercised.  Timings are meaningless and the line says so."""
    import json
    import os
    import subprocess
    import sys
    env = dict(os.environ, ZKP_BENCH_SHARED_GPU="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(H.ROOT, "bench.py"), "--gpus", "2", "--scaling", scaling, "--batch", "66", "--steps", "1", "--warmup", "0", "--cpu-sample", "0",
           "--no-pcie-leg", "--big-batch", "9", "--distinct-batch", "34", "--ck-batch", "1025", "--interactive-batch", "33", "--other-reps", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["verdicts_ok"] and "FUNCTIONAL CHECK ONLY" in d["data"]
    assert d["config"]["proofs_total"] == (132 if scaling == "weak" else 66) and d["config"]["proofs_per_rank"] == (66 if scaling == "weak" else 33)
    legs = d["other_configs"]
    ck = [v for k, v in legs.items() if "configs[3]" in k][0]
    big = [v for k, v in legs.items() if "configs[4]" in k][0]
    assert ck["n_gpus"] == 2 and ck["keys_per_rank"] == 513 and ck["all_rejected_as_expected"]          # 1025 keys: blocks of 513 + 512
    assert big["n_gpus"] == 2 and big["batch_per_rank"] == 5 and big["verdicts_ok"]                       # 9 proofs: blocks of 5 + 4
    assert all(v.get("verdicts_ok", True) and v.get("all_accepted", True) for v in legs.values())


def test_receive_buffers_of_an_8_rank_prove_step_fit_the_gpu():
    """the round-3 verdict: "at N = 8 weak scaling that is a 4.3 GB receive buffer per rank per step that nothing has ever allocated
    on hardware".  shard.GatherBuffers for world = 8 at the headline rank block (4096 proofs, n = 2048): c1 + c2 = 2 x 2.15 GB = 4.29 GB
    next to the rank's own 1.5 GB batch — allocated ONCE here, as bench.make_steps does outside its timed steps."""
    import importlib
    import torch
    shard = importlib.import_module("zk-paillier_amd.shard")
    like = torch.empty((4096, 128, 128), dtype=torch.int32, device="cuda:0")          # one rank's c1: 4096 x 128 rows x 4096 bits
    bufs = [shard.GatherBuffers(like, 8) for _ in range(2)]
    assert all(b.nbytes == 8 * 4096 * 128 * 128 * 4 == 2147483648 for b in bufs) and sum(b.nbytes for b in bufs) == 4294967296 and bufs[0].pad is None
    unequal = shard.GatherBuffers(like[:4093], 8, counts=[4093] * 7 + [4090])
    assert unequal.pad.shape[0] == 4093 and unequal.out.shape[0] == 8 * 4093
    bufs[0].out[-1].fill_(7); torch.cuda.synchronize()
    assert int(bufs[0].out[-1, -1, -1].item()) == 7
    del bufs, unequal, like
    torch.cuda.empty_cache()
