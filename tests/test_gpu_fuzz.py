"""Randomised differential tests: the HIP path against the CPU oracle on random
(N, q, batch, mod factors, in place / out of place) draws, weighted towards the
boundaries between the arithmetic policies and between the launch plans.

The default run draws a few dozen cases; HEXL_AMD_FUZZ_CASES=<n> and
HEXL_AMD_FUZZ_SEED=<s> make it a soak test (`tools/` has no copy of this: only
`tests/` may use the oracle).  Every failure message carries the draw, so a case
can be replayed.
"""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("HEXL_AMD_FUZZ_CASES", "48"))
SEED = int(os.environ.get("HEXL_AMD_FUZZ_SEED", "20260926"))
MAX_ELEMS = 1 << 21  # per case: keeps the scalar oracle below ~0.2 s

# GeneratePrimes bit sizes (q in [2^b, 2^(b+1))) on both sides of the policy boundaries
# (q < 2^30, 2^50, 2^56) and up to the API's limit (q < 2^62)
BITS = [20, 27, 28, 29, 30, 35, 44, 48, 49, 50, 53, 54, 55, 56, 57, 58, 59, 60, 61]


@pytest.fixture(scope="module")
def hx():
    import hexl_amd
    return hexl_amd


@pytest.fixture(scope="module")
def ho():
    from oracle import hexl_oracle
    return hexl_oracle


_prime_cache = {}


def draw_prime(ho, rng, logn):
    """A prime q = 1 mod 2N of a random size; both ends of the size's range are drawn."""
    for _ in range(64):
        bits = rng.choice(BITS)
        if bits < logn + 2:
            continue
        key = (bits, logn)
        if key not in _prime_cache:
            try:
                lo = ho.generate_primes(3, bits, True, 1 << logn)
                hi = ho.generate_primes(3, bits, False, 1 << logn)
                _prime_cache[key] = [int(p) for p in lo] + [int(p) for p in hi]
            except RuntimeError:  # fewer than three such primes in the range
                _prime_cache[key] = []
        if _prime_cache[key]:
            return rng.choice(_prime_cache[key])
    raise AssertionError("no modulus size fits")


def rand_u64(rng, shape, bound):
    """uniform in [0, bound), bound < 2^64, with the extremes planted"""
    g = np.random.default_rng(rng.getrandbits(63))
    x = g.integers(0, bound, size=shape, dtype=np.uint64, endpoint=False)
    flat = x.reshape(-1)
    flat[rng.randrange(flat.size)] = bound - 1
    flat[rng.randrange(flat.size)] = 0
    return x


@pytest.mark.parametrize("idx", range(CASES))
def test_fuzz_ntt(hx, ho, idx):
    rng = random.Random(SEED * 1000003 + idx)
    logn = rng.choice([1, 2, 3, 5, 8, 10, 11, 12, 12, 13, 13, 14, 14, 15, 16, 16, 17])
    n = 1 << logn
    q = draw_prime(ho, rng, logn)
    # batches around the plan thresholds (96 for the one-kernel N = 2^14 plan) and ragged ones
    batch = min(rng.choice([1, 1, 2, 3, 5, 8, 17, 64, 95, 96, 97, 191, 256]), max(1, MAX_ELEMS // n))
    forward = rng.random() < 0.5
    in_mf, out_mf = rng.choice([(1, 1), (2, 1), (4, 1), (1, 4), (2, 4), (4, 4)] if forward
                               else [(1, 1), (2, 1), (1, 2), (2, 2)])
    inplace = rng.random() < 0.5
    tag = dict(idx=idx, n=n, q=q, batch=batch, fwd=forward, in_mf=in_mf, out_mf=out_mf,
               inplace=inplace)
    x = rand_u64(rng, (batch, n), in_mf * q)
    oracle = ho.NTT(n, q)
    exp = (oracle.forward if forward else oracle.inverse)(x, in_mf, out_mf)
    plan = hx.NTT(n, q)
    src = hx.from_numpy(x)
    import torch
    dst = src if inplace else torch.full_like(src, 0x5A5A5A5A)
    (plan.ComputeForward if forward else plan.ComputeInverse)(dst, src, in_mf, out_mf)
    got = hx.to_numpy(dst).reshape(batch, n)
    if out_mf == 1:
        assert np.array_equal(got, exp), tag
    else:
        assert (got < np.uint64(out_mf * q)).all(), tag
        assert np.array_equal(got % np.uint64(q), exp % np.uint64(q)), tag
    if not inplace:
        assert np.array_equal(hx.to_numpy(src).reshape(batch, n), x), tag


_MAPPED = {}


def _mapped_words(hx, words):
    """A pinned, device-mapped region of at least `words` uint64 (hexl_amd_host_alloc), kept for the
    module: (numpy view, base address)."""
    import ctypes as C
    if _MAPPED.get("words", 0) < words:
        if "ptr" in _MAPPED:
            assert hx.lib.hexl_amd_host_free(_MAPPED["ptr"]) == 0
        pm = C.c_void_p()
        assert hx.lib.hexl_amd_host_alloc(C.byref(pm), words * 8) == 0
        _MAPPED.update(ptr=pm, words=words,
                       view=np.ctypeslib.as_array(C.cast(pm, C.POINTER(C.c_uint64)), shape=(words,)))
    return _MAPPED["view"], _MAPPED["ptr"].value


@pytest.mark.parametrize("idx", range(CASES))
def test_fuzz_ntt_host_pointers(hx, ho, idx):
    """The *_host entry points (what intel::hexl::NTT binds for caller memory) on random draws: ordinary or
    pinned device-mapped buffers, in place / out of place, both directions, every legal factor pair, batches
    on both sides of the bounce limit and of the plan thresholds across the link (N = 8192 below four
    polynomials, N = 16384 below 96), the completion flag polled or the stream synchronised -- against the oracle."""
    import ctypes as C
    rng = random.Random(SEED * 104729 + idx)
    logn = rng.choice([2, 6, 10, 11, 12, 12, 13, 13, 13, 14, 14, 14, 15, 16, 16, 17])
    n = 1 << logn
    q = draw_prime(ho, rng, logn)
    batch = min(rng.choice([1, 1, 1, 2, 3, 4, 5, 8, 17, 95, 96, 97]), max(1, (MAX_ELEMS // 2) // n))
    forward = rng.random() < 0.5
    in_mf, out_mf = rng.choice([(1, 1), (2, 1), (4, 1), (1, 4), (2, 4), (4, 4)] if forward
                               else [(1, 1), (2, 1), (1, 2), (2, 2)])
    inplace = rng.random() < 0.5
    mapped = rng.random() < 0.5
    poll = rng.choice([1, 1, 0])
    bounce_kb = rng.choice([512, 512, 64, 0, 2048])
    tag = dict(idx=idx, n=n, q=q, batch=batch, fwd=forward, in_mf=in_mf, out_mf=out_mf, inplace=inplace,
               mapped=mapped, poll=poll, bounce_kb=bounce_kb)
    x = rand_u64(rng, (batch, n), in_mf * q)
    oracle = ho.NTT(n, q)
    exp = (oracle.forward if forward else oracle.inverse)(x, in_mf, out_mf)
    plan = hx.NTT(n, q)
    words = batch * n
    if mapped:
        view, base = _mapped_words(hx, 2 * words)
        src = view[:words]
        src[:] = x.reshape(-1)
        dst = src if inplace else view[words:2 * words]
        p_src = C.c_void_p(base)
        p_dst = p_src if inplace else C.c_void_p(base + words * 8)
    else:
        src = x.reshape(-1).copy()
        dst = src if inplace else np.full(words, 0x5A5A5A5A, dtype=np.uint64)
        p_src, p_dst = src.ctypes.data_as(C.c_void_p), dst.ctypes.data_as(C.c_void_p)
    if not inplace:
        dst[:] = 0x5A5A5A5A
    try:
        hx.set_tuning("host_poll", poll)
        hx.set_tuning("host_bounce_kb", bounce_kb)
        fn = hx.lib.hexl_amd_ntt_forward_host if forward else hx.lib.hexl_amd_ntt_inverse_host
        assert fn(plan._h, p_dst, p_src, batch, in_mf, out_mf) == 0, tag
    finally:
        hx.set_tuning("host_poll", 1)
        hx.set_tuning("host_bounce_kb", 512)
    got = np.array(dst, copy=True).reshape(batch, n)
    if out_mf == 1:
        assert np.array_equal(got, exp), tag
    else:
        assert (got < np.uint64(out_mf * q)).all(), tag
        assert np.array_equal(got % np.uint64(q), exp % np.uint64(q)), tag
    if not inplace:
        assert np.array_equal(np.array(src).reshape(batch, n), x), tag
    assert hx.get_counter("host_poll_timeouts") == 0, tag


@pytest.mark.parametrize("idx", range(CASES))
def test_fuzz_ntt_rns(hx, ho, idx):
    """Several moduli in one call (the multi-plan launches), mixed policies allowed."""
    rng = random.Random(SEED * 7919 + idx)
    logn = rng.choice([3, 8, 10, 12, 13, 14, 15, 16])
    n = 1 << logn
    primes = []
    while len(primes) < rng.choice([2, 3, 5, 8]):
        p = draw_prime(ho, rng, logn)
        if p not in primes:
            primes.append(p)
    polys = min(rng.choice([1, 2, 3, 7]), max(1, MAX_ELEMS // (n * len(primes))))
    forward = rng.random() < 0.5
    tag = dict(idx=idx, n=n, primes=primes, polys=polys, fwd=forward)
    x = np.stack([rand_u64(rng, (polys, n), q) for q in primes])  # (prime, poly, n)
    exp = np.stack([(ho.NTT(n, q).forward if forward else ho.NTT(n, q).inverse)(x[i], 1, 1)
                    for i, q in enumerate(primes)])
    plans = [hx.NTT(n, q) for q in primes]
    buf = hx.from_numpy(x.reshape(-1)).reshape(len(primes), polys, n)
    (hx.ComputeForwardRNS if forward else hx.ComputeInverseRNS)(plans, buf, buf, 1, 1)
    assert np.array_equal(hx.to_numpy(buf).reshape(x.shape), exp), tag


@pytest.mark.parametrize("idx", range(CASES))
def test_fuzz_ntt_map(hx, ho, idx):
    """The per-polynomial prime map (hexl_amd_ntt_*_map / _indexed): random tables (repeats,
    permutations), `inner`, periods, ragged ends, mixed arithmetic policies, lazy outputs."""
    import torch
    rng = random.Random(SEED * 15485863 + idx)
    logn = rng.choice([3, 10, 12, 12, 13, 14, 15, 16, 18])
    n = 1 << logn
    primes = []
    while len(primes) < rng.choice([1, 2, 3, 5, 8]):
        p = draw_prime(ho, rng, logn)
        if p not in primes:
            primes.append(p)
    period = rng.choice([1, 2, 3, 4, 7, 8])
    tab = [rng.randrange(len(primes)) for _ in range(period)]
    inner = rng.choice([1, 1, 1, 2, 3, 5])
    polys = min(rng.choice([1, 2, 5, 8, 13, 24, 31]), max(1, MAX_ELEMS // n))
    which = [tab[(i // inner) % period] for i in range(polys)]
    forward = rng.random() < 0.5
    in_mf, out_mf = rng.choice([(1, 1), (2, 1), (4, 4), (1, 4)] if forward
                               else [(1, 1), (2, 1), (2, 2)])
    indexed = rng.random() < 0.4
    tag = dict(idx=idx, n=n, primes=primes, tab=tab, inner=inner, polys=polys, fwd=forward,
               in_mf=in_mf, out_mf=out_mf, indexed=indexed)
    x = np.stack([rand_u64(rng, (n,), in_mf * primes[k]) for k in which])
    onts = [ho.NTT(n, q) for q in primes]
    exp = np.stack([(onts[k].forward if forward else onts[k].inverse)(x[i], in_mf, out_mf)
                    for i, k in enumerate(which)])
    plans = [hx.NTT(n, q) for q in primes]
    src = hx.from_numpy(x)
    dst = src if rng.random() < 0.5 else torch.full_like(src, 0x3C3C3C3C)
    if indexed:
        (hx.ComputeForwardIndexed if forward else hx.ComputeInverseIndexed)(
            plans, which, dst, src, in_mf, out_mf)
    else:
        (hx.ComputeForwardMap if forward else hx.ComputeInverseMap)(
            plans, tab, inner, dst, src, in_mf, out_mf)
    got = hx.to_numpy(dst).reshape(polys, n)
    for i, k in enumerate(which):
        q = np.uint64(primes[k])
        if out_mf == 1:
            assert np.array_equal(got[i], exp[i]), (tag, i)
        else:
            assert (got[i] < np.uint64(out_mf) * q).all(), (tag, i)
            assert np.array_equal(got[i] % q, exp[i] % q), (tag, i)


ELT_OPS = ["add", "add_scalar", "sub", "sub_scalar", "mult", "fma", "fma_null", "reduce",
           "cmp_add", "cmp_sub_mod"]


@pytest.mark.parametrize("idx", range(CASES))
def test_fuzz_eltwise(hx, ho, idx):
    rng = random.Random(SEED * 104729 + idx)
    op = rng.choice(ELT_OPS)
    n = rng.choice([1, 2, 7, 8, 9, 63, 64, 65, 255, 1000, 4096, 4099, 65536, 100003, 1 << 20])
    import torch
    tag = dict(idx=idx, op=op, n=n)

    def prime(max_bits):
        bits = rng.choice([b for b in BITS if b <= max_bits])
        return int(ho.generate_primes(1, bits, rng.random() < 0.5, 1)[0])

    if op in ("add", "sub", "add_scalar", "sub_scalar"):
        q = prime(61)
        a = rand_u64(rng, n, q)
        if op.endswith("scalar"):
            s = rng.randrange(q)
            exp = (ho.eltwise_add_mod if op[0] == "a" else ho.eltwise_sub_mod)(a, s, q)
            r = hx.from_numpy(a)
            (hx.EltwiseAddMod if op[0] == "a" else hx.EltwiseSubMod)(r, r, s, n, q)
        else:
            b = rand_u64(rng, n, q)
            exp = (ho.eltwise_add_mod if op[0] == "a" else ho.eltwise_sub_mod)(a, b, q)
            r = torch.zeros(n, dtype=torch.int64, device="cuda")
            (hx.EltwiseAddMod if op[0] == "a" else hx.EltwiseSubMod)(
                r, hx.from_numpy(a), hx.from_numpy(b), n, q)
    elif op == "mult":
        in_mf = rng.choice([1, 2, 4])
        q = prime(61 if in_mf <= 2 else 60)
        a, b = rand_u64(rng, n, in_mf * q), rand_u64(rng, n, in_mf * q)
        exp = ho.eltwise_mult_mod(a, b, q, in_mf)
        r = hx.from_numpy(a)
        hx.EltwiseMultMod(r, r, hx.from_numpy(b), n, q, in_mf)
    elif op in ("fma", "fma_null"):
        in_mf = rng.choice([1, 2, 4, 8])
        q = prime(60)
        a = rand_u64(rng, n, in_mf * q)
        s = rng.randrange(in_mf * q)
        c = rand_u64(rng, n, in_mf * q) if op == "fma" else None
        exp = ho.eltwise_fma_mod(a, s, c, q, in_mf)
        r = torch.zeros(n, dtype=torch.int64, device="cuda")
        hx.EltwiseFMAMod(r, hx.from_numpy(a), s, hx.from_numpy(c) if c is not None else None,
                         n, q, in_mf)
    elif op == "reduce":
        q = prime(61)
        in_mf, out_mf = rng.choice([(q, 1), (q, 2), (2, 1), (4, 1), (4, 2)])
        bound = (1 << 64) if in_mf == q else in_mf * q
        a = rand_u64(rng, n, bound)
        exp = ho.eltwise_reduce_mod(a, q, in_mf, out_mf)
        r = torch.zeros(n, dtype=torch.int64, device="cuda")
        hx.EltwiseReduceMod(r, hx.from_numpy(a), n, q, in_mf, out_mf)
        tag.update(q=q, in_mf=in_mf, out_mf=out_mf)
        got = hx.to_numpy(r)
        if in_mf == q and out_mf == 2:  # the one lazy output of the family
            assert (got < np.uint64(2 * q)).all(), tag
            assert np.array_equal(got % np.uint64(q), exp % np.uint64(q)), tag
            return
    elif op == "cmp_add":
        cmp = rng.randrange(8)
        bound, diff = rng.getrandbits(rng.choice([3, 20, 63])), rng.getrandbits(40) + 1
        a = rand_u64(rng, n, 1 << rng.choice([4, 21, 62]))
        exp = ho.eltwise_cmp_add(a, cmp, bound, diff)
        r = hx.from_numpy(a)
        hx.EltwiseCmpAdd(r, r, n, cmp, bound, diff)
        tag.update(cmp=cmp, bound=bound, diff=diff)
    else:
        cmp = rng.randrange(8)
        q = prime(61)
        bound, diff = rng.randrange(1 << 63), rng.randrange(1, q)
        a = rand_u64(rng, n, 1 << 64)
        exp = ho.eltwise_cmp_sub_mod(a, q, cmp, bound, diff)
        r = hx.from_numpy(a)
        hx.EltwiseCmpSubMod(r, r, n, q, cmp, bound, diff)
        tag.update(q=q, cmp=cmp, bound=bound, diff=diff)
    assert np.array_equal(hx.to_numpy(r), np.asarray(exp, dtype=np.uint64)), tag


@pytest.mark.parametrize("idx", range(max(8, CASES // 4)))
def test_fuzz_key_switch(hx, ho, idx):
    """KeySwitchBatch with random sizes and RNS bases of mixed modulus size (hence mixed
    arithmetic policy inside one multi-plan launch) against the oracle, target by target."""
    rng = random.Random(SEED * 31337 + idx)
    logn = rng.choice([3, 6, 10, 11, 12, 13, 14])
    n = 1 << logn
    D = rng.choice([1, 2, 3, 4, 7])
    K = D + 1 + rng.choice([0, 0, 1, 2])
    C = rng.choice([2, 2, 3])
    T = rng.choice([1, 2, 3, 5]) if n * D * K <= (1 << 18) else 1
    moduli = []
    while len(moduli) < K:
        bits = rng.choice([b for b in BITS if logn + 3 <= b <= 59])  # SEAL's limit: 60-bit primes
        try:
            p = int(ho.generate_primes(1, bits, rng.random() < 0.5, n)[0])
        except RuntimeError:
            continue
        if p not in moduli:
            moduli.append(p)
    tag = dict(idx=idx, n=n, D=D, K=K, C=C, T=T, moduli=moduli)
    keys = [np.concatenate([rand_u64(rng, n, moduli[i]) for _ in range(C) for i in range(K)])
            for _ in range(D)]
    msf = [rng.randrange(1, moduli[i]) for i in range(D)]
    targets = [np.concatenate([rand_u64(rng, n, moduli[j]) for j in range(D)]) for _ in range(T)]
    results = [np.concatenate([rand_u64(rng, n, moduli[i]) for _ in range(C) for i in range(D)])
               for _ in range(T)]
    want = np.concatenate([ho.key_switch(results[t], targets[t], n, D, K, D + 1, C, moduli, keys, msf)
                           for t in range(T)])
    d_res = hx.from_numpy(np.concatenate(results))
    hx.KeySwitchBatch(d_res, hx.from_numpy(np.concatenate(targets)), T, n, D, K, D + 1, C, moduli,
                      [hx.from_numpy(k) for k in keys], msf)
    assert np.array_equal(hx.to_numpy(d_res), want), tag


@pytest.mark.parametrize("idx", range(max(8, CASES // 4)))
def test_fuzz_dyadic_multiply(hx, ho, idx):
    rng = random.Random(SEED * 2741 + idx)
    n = rng.choice([8, 64, 1000, 1024, 4096, 8192, 16384])
    k = rng.choice([1, 2, 3, 5, 9])
    pairs = rng.choice([1, 2, 4]) if n * k <= (1 << 16) else 1
    moduli = [int(ho.generate_primes(1, rng.choice([b for b in BITS if b <= 60]),
                                     rng.random() < 0.5, 1)[0]) for _ in range(k)]
    tag = dict(idx=idx, n=n, moduli=moduli, pairs=pairs)
    ops1 = [np.concatenate([rand_u64(rng, n, q) for _ in range(2) for q in moduli]) for _ in range(pairs)]
    ops2 = [np.concatenate([rand_u64(rng, n, q) for _ in range(2) for q in moduli]) for _ in range(pairs)]
    want = np.concatenate([ho.dyadic_multiply(ops1[p], ops2[p], n, moduli) for p in range(pairs)])
    import torch
    d_res = torch.zeros(3 * n * k * pairs, dtype=torch.int64, device="cuda")
    hx.DyadicMultiplyBatch(d_res, hx.from_numpy(np.concatenate(ops1)), hx.from_numpy(np.concatenate(ops2)),
                           pairs, n, moduli)
    assert np.array_equal(hx.to_numpy(d_res), want), tag
