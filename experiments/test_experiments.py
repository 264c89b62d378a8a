"""Parity tests of the plans that were built, measured and NOT adopted (fused_pass: both passes
in one persistent launch; mixed launches over chunks; the recursive sequence lock under the fused
plan), lifted out of tests/test_gpu_parity.py in round 4 for the record.  They need a library
built from the archived sources (experiments/README.md: the tree at commit 55a090c with
-DHEXL_AMD_EXPERIMENTS) and are not part of `pytest tests/` -- the product has no such plan to
select.  Marker: experiments.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.experiments
PLAN_FUSED, PLAN_SPLIT, PLAN_TILED, PLAN_MIXED = 0, 1, 2, 3


@pytest.fixture(scope="module")
def hx():
    import hexl_amd
    if hexl_amd.lib.hexl_amd_set_tuning(b"experiments", 1) != 0:
        pytest.skip("the loaded library is not an experiments build")
    for name, value in (("PLAN_FUSED", 0), ("PLAN_SPLIT", 1), ("PLAN_TILED", 2), ("PLAN_MIXED", 3)):
        setattr(hexl_amd, name, value)
    return hexl_amd


@pytest.fixture(scope="module")
def ho():
    from oracle import hexl_oracle
    return hexl_oracle


def dev(hx, a):
    return hx.from_numpy(a)


def host(hx, t):
    return hx.to_numpy(t)


@pytest.mark.parametrize("logn", [15, 16])
@pytest.mark.parametrize("bits", [28, 45, 54, 60])
def test_ntt_fused_plan_matches_split_plan(hx, logn, bits):
    """The one-launch plan (fused_pass: persistent workgroups, per-XCD tickets, the
    intermediate handed from the strided phase to the tile phase through the XCD's L2)
    gives the same bits as the default two-launch plan -- canonical and lazy outputs,
    in place and out of place, batches that do and do not fill the chip."""
    import torch
    n = 1 << logn
    q = hx.GeneratePrimes(1, bits, True, n)[0]
    ntt = hx.NTT(n, q)
    try:
        hx.set_tuning("fused_min_batch", 1)
        for batch in (1, 67, 640):
            x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
            hx.fill_splitmix(x, n, batch, 11 + logn, q)
            for fwd in (True, False):
                fn = ntt.ComputeForward if fwd else ntt.ComputeInverse
                for out_mf in ((1, 4) if fwd else (1, 2)):
                    res = {}
                    for plan in (hx.PLAN_SPLIT, hx.PLAN_FUSED):
                        hx.set_tuning("plan", plan)
                        a = x.clone()
                        fn(a, a, 1, out_mf)
                        b = torch.full_like(x, -1)
                        fn(b, x, 1, out_mf)
                        assert torch.equal(a, b)
                        res[plan] = a
                    assert torch.equal(res[hx.PLAN_SPLIT], res[hx.PLAN_FUSED])
    finally:
        hx.set_tuning("plan", hx.PLAN_SPLIT)
        hx.set_tuning("fused_min_batch", 64)


@pytest.mark.parametrize("bits", [54, 49])
def test_ntt_mixed_plan_matches_split_plan(hx, bits):
    """The mixed plan (N = 2^16: workgroups of chunk i's first pass and of chunk i-1's
    second pass in one launch) gives the same bits as the default two-launch plan -- full,
    ragged and single-chunk pipelines, canonical and lazy outputs, in place and out of place."""
    import torch
    n = 65536
    q = hx.GeneratePrimes(1, bits, True, n)[0]
    ntt = hx.NTT(n, q)
    try:
        for chunk, batch in ((4, 8), (4, 11), (3, 7), (16, 64)):
            hx.set_tuning("mixed_chunk", chunk)
            x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
            hx.fill_splitmix(x, n, batch, 5 + chunk, q)
            for fwd in (True, False):
                fn = ntt.ComputeForward if fwd else ntt.ComputeInverse
                for out_mf in ((1, 4) if fwd else (1, 2)):
                    res = {}
                    for plan in (hx.PLAN_SPLIT, hx.PLAN_MIXED):
                        hx.set_tuning("plan", plan)
                        a = x.clone()
                        fn(a, a, 1, out_mf)
                        b = torch.full_like(x, -1)
                        fn(b, x, 1, out_mf)
                        assert torch.equal(a, b)
                        res[plan] = a
                    got, want = res[hx.PLAN_MIXED], res[hx.PLAN_SPLIT]
                    if out_mf == 1:
                        assert torch.equal(got, want)
                    else:  # lazy outputs: same residues, inside the reference's range
                        assert int(got.min()) >= 0 and int(got.max()) < out_mf * q
                        assert torch.equal(got % q, want % q)
    finally:
        hx.set_tuning("plan", hx.PLAN_SPLIT)
        hx.set_tuning("mixed_chunk", 512)


def test_key_switch_batch_under_the_fused_plan(hx, ho):
    """Round-2 advisor finding: KeySwitch holds its stream's sequence lock while it enqueues and
    its inverse transforms (T * C >= fused_min_batch polynomials of N = 32768) took the same
    lock again inside the one-launch fused plan -- a self-deadlock on a non-recursive mutex.
    The lock is recursive now; the call must return, bit-exact.  (Experiments builds only:
    the default build has no fused plan to select.)"""
    import threading
    n, D, K, C, T = 32768, 2, 3, 2, 33  # T * C = 66 >= 64
    rng = np.random.default_rng(5)
    moduli = [int(q) for q in ho.generate_primes(K, 54, True, n)]
    keys = [np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                            for _ in range(C) for i in range(K)]) for _ in range(D)]
    msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
    targets = [np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64) for j in range(D)])
               for _ in range(T)]
    results = [np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                               for _ in range(C) for i in range(D)]) for _ in range(T)]
    d_keys = [dev(hx, k) for k in keys]
    d_t = dev(hx, np.concatenate(targets))
    want = dev(hx, np.concatenate(results))
    hx.KeySwitchBatch(want, d_t, T, n, D, K, D + 1, C, moduli, d_keys, msf)  # split plan
    check = ho.key_switch(results[0], targets[0], n, D, K, D + 1, C, moduli, keys, msf)
    assert np.array_equal(host(hx, want)[:check.size], check)
    got = dev(hx, np.concatenate(results))
    done = []

    def run():
        hx.KeySwitchBatch(got, d_t, T, n, D, K, D + 1, C, moduli, d_keys, msf)
        done.append(True)

    try:
        hx.set_tuning("plan", hx.PLAN_FUSED)
        th = threading.Thread(target=run, daemon=True)
        th.start()
        th.join(120)
        assert done, "KeySwitchBatch under the fused plan did not return (sequence lock)"
    finally:
        hx.set_tuning("plan", hx.PLAN_SPLIT)
    assert np.array_equal(host(hx, got), host(hx, want))


