"""GPU parity tests: the HIP path (through the C-ABI, via the hexl_amd Python
mirror of intel::hexl::NTT / Eltwise*) against the CPU oracle and the
reference's known-answer vectors.  Bit-exact for canonical outputs; lazy
outputs are compared modulo q plus a range check, exactly as the reference's
own tests do (test/test-ntt.cpp:246-251, test/test-ntt-avx512.cpp:268-278).
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "hexl_kat.json")))


@pytest.fixture(scope="module")
def hx():
    import hexl_amd
    return hexl_amd


@pytest.fixture(scope="module")
def ho():
    from oracle import hexl_oracle
    return hexl_oracle


def U(x):
    return np.asarray(x, dtype=np.uint64)


def dev(hx, a):
    return hx.from_numpy(U(a))


def host(hx, t):
    return hx.to_numpy(t)


def resolve_q(ho, q):
    return ho.generate_primes(*q["gp"])[0] if isinstance(q, dict) else q


def resolve(v, q):
    if isinstance(v, dict):
        return q - v["q_minus"]
    if isinstance(v, list):
        return [resolve(x, q) for x in v]
    return v


# ---------------------------------------------------------------- NTT KATs
@pytest.mark.parametrize("case", KAT["ntt_forward"]["cases"],
                         ids=lambda c: f"n{c['n']}_q{c['q']}")
def test_ntt_kat_api(hx, case):
    """TEST_P(DegreeModulusInputOutput, API), test/test-ntt.cpp:227-339."""
    import torch
    n, q = case["n"], case["q"]
    inp, exp = U(case["in"]), U(case["out"])
    ntt = hx.NTT(n, q)
    # in-place forward
    buf = dev(hx, inp)
    ntt.ComputeForward(buf, buf, 1, 1)
    assert (host(hx, buf) == exp).all()
    # in-place lazy forward
    buf = dev(hx, inp)
    ntt.ComputeForward(buf, buf, 2, 4)
    got = host(hx, buf)
    assert (got < 4 * q).all() and (got % np.uint64(q) == exp).all()
    # out-of-place round trip; the destination starts as garbage (99)
    src = dev(hx, inp)
    out = torch.full_like(src, 99)
    ntt.ComputeForward(out, src, 1, 1)
    assert (host(hx, out) == exp).all()
    assert (host(hx, src) == inp).all()
    back = torch.full_like(src, 99)
    ntt.ComputeInverse(back, out, 1, 1)
    assert (host(hx, back) == inp).all()
    # forward with in_mf = 2
    ntt.ComputeForward(out, src, 2, 1)
    assert (host(hx, out) == exp).all()
    # lazy inverse
    ntt.ComputeInverse(back, out, 1, 2)
    got = host(hx, back)
    assert (got < 2 * q).all() and (got % np.uint64(q) == inp).all()
    # in-place inverse
    ntt.ComputeInverse(out, out, 1, 1)
    assert (host(hx, out) == inp).all()


@pytest.mark.parametrize("case", KAT["ntt_root_powers"]["cases"])
def test_ntt_root_powers(hx, case):
    ntt = hx.NTT(case["n"], case["q"])
    assert [int(x) for x in ntt.GetRootOfUnityPowers()] == case["powers"]


def test_ntt_tables_match_oracle(hx, ho):
    n, q = 1024, 0xffffee001
    a, b = hx.NTT(n, q), ho.NTT(n, q)
    assert a.GetMinimalRootOfUnity() == b.w == 46310425
    assert (a.GetRootOfUnityPowers() == b.root_pows).all()
    assert (a.GetPrecon64RootOfUnityPowers() == b.precon_root_pows).all()
    assert (a.GetInvRootOfUnityPowers() == b.inv_root_pows).all()
    assert (a.GetPrecon64InvRootOfUnityPowers() == b.precon_inv_root_pows).all()
    assert [int(x) for x in a.GetPrecon32RootOfUnityPowers()] == [
        (int(w) << 32) // q for w in b.root_pows]
    assert [int(x) for x in a.GetPrecon52InvRootOfUnityPowers()] == [
        (int(w) << 52) // q for w in b.inv_root_pows]


def test_ntt_custom_root(hx, ho):
    """NTT(N, q, root): test/test-ntt.cpp:200-216 uses the minimal root; also
    check a non-minimal primitive root against the oracle."""
    n, q = 8, 769
    w = ho.minimal_primitive_root(2 * n, q)
    x = dev(hx, [1, 2, 3, 4, 5, 6, 7, 8])
    y1, y2 = x.clone(), x.clone()
    hx.NTT(n, q).ComputeForward(y1, y1, 1, 1)
    hx.NTT(n, q, w).ComputeForward(y2, y2, 1, 1)
    assert (host(hx, y1) == host(hx, y2)).all()
    w3 = pow(w, 3, q)
    y3 = x.clone()
    hx.NTT(n, q, w3).ComputeForward(y3, y3, 1, 1)
    assert (host(hx, y3) == ho.NTT(n, q, w3).forward(host(hx, x), 1, 1)).all()


# ---------------------------------------------------------------- NTT vs oracle
SWEEP_N = [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384,
           32768, 65536, 131072]


@pytest.mark.parametrize("n", SWEEP_N)
@pytest.mark.parametrize("bits", [27, 33, 49, 54, 56, 58, 60, 61])
def test_ntt_vs_oracle(hx, ho, n, bits):
    """Random inputs, all legal (in_mf, out_mf), in-place and out-of-place,
    small and large moduli (pattern of test/test-ntt-avx512.cpp:169-398 and
    test/test-ntt.cpp:406-478): one bit size per arithmetic policy -- Small, Fp64 (33, 49), Lazy,
    Lazy32 (just above 2^56), Lazy16 (just above 2^58), Harvey60, Strict."""
    import torch
    q = ho.generate_primes(1, bits, bits % 2 == 0, n)[0]
    batch = 3 if n <= 16384 else 2
    if n > 8192 and bits in (27, 33, 56, 58, 60):
        batch = 1  # keeps the oracle's share of the sweep small
    ont, gnt = ho.NTT(n, q), hx.NTT(n, q)
    x = np.stack([ho.fill_splitmix(n, bits * 1000 + n + b, q) for b in range(batch)])
    f_ref = ont.forward(x, 1, 1)
    dx = dev(hx, x)
    out = torch.empty_like(dx)
    gnt.ComputeForward(out, dx, 1, 1)
    assert (host(hx, out) == f_ref).all()
    assert (host(hx, dx) == x).all()  # operand untouched out-of-place
    inplace = dx.clone()
    gnt.ComputeForward(inplace, inplace, 1, 1)
    assert (host(hx, inplace) == f_ref).all()
    # inverse, canonical, out-of-place and in-place
    back = torch.empty_like(dx)
    gnt.ComputeInverse(back, out, 1, 1)
    assert (host(hx, back) == x).all()
    gnt.ComputeInverse(out, out, 1, 1)
    assert (host(hx, out) == x).all()
    # lazy ranges
    for in_mf, out_mf in ((1, 4), (2, 1), (2, 4), (4, 1), (4, 4)):
        xin = np.stack([ho.fill_splitmix(n, 77 + in_mf + b, in_mf * q) for b in range(batch)])
        ref = ont.forward(xin % np.uint64(q), 1, 1)
        d = dev(hx, xin)
        gnt.ComputeForward(d, d, in_mf, out_mf)
        got = host(hx, d)
        assert (got < out_mf * q).all()
        assert ((got % np.uint64(q)) == ref).all()
        if out_mf == 1:
            assert (got == ref).all()
    for in_mf, out_mf in ((1, 2), (2, 1), (2, 2)):
        xin = np.stack([ho.fill_splitmix(n, 99 + in_mf + b, in_mf * q) for b in range(batch)])
        ref = ont.inverse(xin % np.uint64(q), 1, 1)
        d = dev(hx, xin)
        gnt.ComputeInverse(d, d, in_mf, out_mf)
        got = host(hx, d)
        assert (got < out_mf * q).all()
        assert ((got % np.uint64(q)) == ref).all()
        if out_mf == 1:
            assert (got == ref).all()


def test_ntt_ragged_batches(hx, ho):
    """Batches that do not fill a 4096-element tile, or straddle one."""
    import torch
    for n, batch in ((2, 1), (2, 5), (64, 1), (64, 65), (1024, 3), (1024, 5), (4096, 1)):
        q = ho.generate_primes(1, 54, True, n)[0]
        x = np.stack([ho.fill_splitmix(n, 5 + b, q) for b in range(batch)])
        ont, gnt = ho.NTT(n, q), hx.NTT(n, q)
        d = dev(hx, x)
        guard = torch.full((d.numel() + 8192,), -7, dtype=torch.int64, device="cuda")
        res = guard[4096:4096 + d.numel()].view_as(d)
        gnt.ComputeForward(res, d, 1, 1)
        assert (host(hx, res) == ont.forward(x, 1, 1)).all()
        assert (guard[:4096] == -7).all() and (guard[4096 + d.numel():] == -7).all()
        gnt.ComputeInverse(res, res, 1, 1)
        assert (host(hx, res) == x).all()
        assert (guard[:4096] == -7).all() and (guard[4096 + d.numel():] == -7).all()


def test_ntt_misaligned_views(hx, ho):
    """Buffers that are only 8-byte aligned (the reference's API promises no
    more than uint64_t alignment)."""
    import torch
    for n in (64, 4096, 65536):
        q = ho.generate_primes(1, 54, True, n)[0]
        x = np.stack([ho.fill_splitmix(n, 3 + b, q) for b in range(2)])
        ont, gnt = ho.NTT(n, q), hx.NTT(n, q)
        raw_in = torch.zeros(2 * n + 3, dtype=torch.int64, device="cuda")
        raw_out = torch.zeros(2 * n + 3, dtype=torch.int64, device="cuda")
        vin, vout = raw_in[1:1 + 2 * n], raw_out[1:1 + 2 * n]
        assert vin.data_ptr() % 16 == 8
        vin.copy_(hx.from_numpy(x).reshape(-1))
        gnt.ComputeForward(vout, vin, 1, 1)
        assert (host(hx, vout).reshape(2, n) == ont.forward(x, 1, 1)).all()
        gnt.ComputeInverse(vout, vout, 1, 1)
        assert (host(hx, vout).reshape(2, n) == x).all()
        assert int(raw_out[0]) == 0 and int(raw_out[-1]) == 0


def test_ntt_zeros(hx):
    """NttNativeTest.ForwardZeros / InverseZeros, test/test-ntt.cpp:406-418."""
    import torch
    for n in (16, 1024, 65536):
        q = hx.GeneratePrimes(1, 54, True, n)[0]
        ntt = hx.NTT(n, q)
        z = torch.zeros(n, dtype=torch.int64, device="cuda")
        ntt.ComputeForward(z, z, 1, 1)
        assert int(z.abs().sum()) == 0
        ntt.ComputeInverse(z, z, 1, 1)
        assert int(z.abs().sum()) == 0


def test_ntt_argument_errors(hx):
    """The reference throws only under HEXL_DEBUG (test/test-ntt.cpp:21-94);
    the C-ABI reports the same contract violations in all builds."""
    import torch
    with pytest.raises(hx.HexlAmdError):
        hx.NTT(1024, 0xffffee001 + 2)  # not prime / not 1 mod 2N
    with pytest.raises(hx.HexlAmdError):
        hx.NTT(1000, 0xffffee001)  # not a power of two
    with pytest.raises(hx.HexlAmdError):
        hx.NTT(8, 769, 2)  # not a primitive 2N-th root
    ntt = hx.NTT(8, 769)
    x = torch.zeros(8, dtype=torch.int64, device="cuda")
    for bad in ((3, 1), (1, 2), (8, 1)):
        with pytest.raises(hx.HexlAmdError):
            ntt.ComputeForward(x, x, *bad)
    for bad in ((4, 1), (1, 4)):
        with pytest.raises(hx.HexlAmdError):
            ntt.ComputeInverse(x, x, *bad)
    with pytest.raises(hx.HexlAmdError):
        ntt.ComputeForward(x, x.cpu(), 1, 1)


def test_ntt_rns(hx, ho):
    """BASELINE configs[3] shape at reduced batch: 8 RNS primes x B polys."""
    import torch
    n, B = 65536, 2
    primes = KAT["generate_primes_survey_probe"]["cases"][1]["out"]
    plans = [hx.NTT(n, p) for p in primes]
    x = np.stack([np.stack([ho.fill_splitmix(n, 1000 * k + b, p) for b in range(B)])
                  for k, p in enumerate(primes)])
    d = dev(hx, x)
    out = torch.empty_like(d)
    hx.ComputeForwardRNS(plans, out, d, 1, 1)
    got = host(hx, out)
    for k, p in enumerate(primes):
        assert (got[k] == ho.NTT(n, p).forward(x[k], 1, 1)).all()
    hx.ComputeInverseRNS(plans, out, out, 1, 1)
    assert (host(hx, out) == x).all()


@pytest.mark.parametrize("n,bits_list,B", [
    (4096, [54, 54, 54], 3),          # one launch sequence over the three moduli
    (8192, [45, 45, 45, 45], 1),      # Fp64 policy, one polynomial per modulus
    (1 << 17, [60, 60], 2),           # Harvey60 policy (just above 2^60), 5 + 12 stages
    (1 << 16, [61, 61], 2),           # Strict policy
    (1 << 15, ["s8", "s8", 54], 2),   # Strict policy just below 2^61 beside Lazy: multi-plan kernels
    (2048, [54, 54], 3),              # below the multi-plan shapes: plan by plan
    (16384, [28, 54, 45, 60], 2),     # mixed arithmetic policies: plan by plan
    (4096, [54] * 33, 1),             # more moduli than one launch takes (32): two groups
])
def test_ntt_rns_shapes(hx, ho, n, bits_list, B):
    """The RNS entry point (polynomials [k*B, (k+1)*B) use plans[k]) over the shapes that do
    and do not go through the multi-plan launches, against the oracle."""
    import torch
    primes = []
    for bits in sorted(set(bits_list), key=str):
        if bits == "s8":  # walking down from 2^61
            found = ho.generate_primes(bits_list.count(bits), 60, False, n)
        else:
            found = ho.generate_primes(bits_list.count(bits), bits, True, n)
        primes += found
    primes = [primes.pop(0) for _ in bits_list]  # any order will do; all distinct
    plans = [hx.NTT(n, p) for p in primes]
    x = np.stack([np.stack([ho.fill_splitmix(n, 100 * k + b, p) for b in range(B)])
                  for k, p in enumerate(primes)])
    d = dev(hx, x)
    out = torch.empty_like(d)
    hx.ComputeForwardRNS(plans, out, d, 1, 1)
    got = host(hx, out)
    for k, p in enumerate(primes):
        assert (got[k] == ho.NTT(n, p).forward(x[k], 1, 1)).all(), (k, p)
    hx.ComputeInverseRNS(plans, out, out, 1, 1)
    assert (host(hx, out) == x).all()


def _mixed_primes(ho, n, bits_list):
    primes = []
    pools = {b: ho.generate_primes(bits_list.count(b), b, True, n) for b in set(bits_list)}
    for b in bits_list:
        primes.append(pools[b].pop(0))
    return primes


@pytest.mark.parametrize("n,bits_list,period_tab,inner,polys", [
    # SEAL's [ciphertext][component][modulus][N]: modulus j at polynomial index ... * k + j
    (4096, [54, 54], [0, 1], 1, 12),
    (8192, [54, 45, 54], [0, 1, 2], 1, 9),                 # period 3, Lazy + Fp64 mixed
    (65536, [54] * 8, list(range(8)), 1, 16),              # configs[3]'s 8 primes, interleaved
    (16384, [28, 54, 45, 60, 61, 54, 45, 59], list(range(8)), 1, 24),  # five policies
    (65536, [56, 54, 58, 57, 59], list(range(5)), 1, 10),              # the Lazy family + Harvey60
    (8192, [43, 49, 36, 54], list(range(4)), 1, 12),                   # Fp64L + Fp64 + Lazy in one call
    (4096, [54, 49, 60], [0, 1, 2, 1], 2, 19),             # inner 2, a plan used twice, ragged end
    (32768, [54, 54, 54], [2, 0, 1], 3, 10),               # permuted table, last slot cut short
    (2048, [54, 45], [0, 1], 1, 6),                        # below the multi-plan degrees: run by run
    (1 << 18, [54, 60], [0, 1], 1, 4),                     # above them: run by run
])
def test_ntt_map_layouts(hx, ho, n, bits_list, period_tab, inner, polys):
    """The per-polynomial prime map (hexl/experimental/seal/key-switch-internal.cpp:60-90
    indexes t_target[j * coeff_count] per modulus j inside one ciphertext component): polynomial
    i uses plans[tab[(i // inner) % period]], one call over interleaved layouts, against the
    oracle -- forward canonical, forward lazy (range + congruence), inverse round trip."""
    import torch
    primes = _mixed_primes(ho, n, bits_list)
    plans = [hx.NTT(n, p) for p in primes]
    onts = [ho.NTT(n, p) for p in primes]
    which = [period_tab[(i // inner) % len(period_tab)] for i in range(polys)]
    x = np.stack([ho.fill_splitmix(n, 7000 + i, primes[k]) for i, k in enumerate(which)])
    d = dev(hx, x)
    out = torch.empty_like(d)
    hx.ComputeForwardMap(plans, period_tab, inner, out, d, 1, 1)
    got = host(hx, out)
    want = np.stack([onts[k].forward(x[i], 1, 1) for i, k in enumerate(which)])
    assert (got == want).all(), [i for i in range(polys) if not (got[i] == want[i]).all()]
    lazy = torch.empty_like(d)
    hx.ComputeForwardMap(plans, period_tab, inner, lazy, d, 1, 4)
    lz = host(hx, lazy)
    for i, k in enumerate(which):
        q = np.uint64(primes[k])
        assert (lz[i] < np.uint64(4) * q).all() and (lz[i] % q == want[i]).all(), i
    hx.ComputeInverseMap(plans, period_tab, inner, out, out, 1, 1)  # in place
    assert (host(hx, out) == x).all()
    # the explicit per-polynomial index recovers the same structure
    out2 = torch.empty_like(d)
    hx.ComputeForwardIndexed(plans, which, out2, d, 1, 1)
    assert (host(hx, out2) == want).all()
    hx.ComputeInverseIndexed(plans, which, out2, out2, 1, 1)
    assert (host(hx, out2) == x).all()


def test_ntt_indexed_without_period(hx, ho):
    """An index array with no periodic structure is served run by run."""
    import torch
    n = 4096
    primes = _mixed_primes(ho, n, [54, 45, 60])
    plans = [hx.NTT(n, p) for p in primes]
    which = [0, 0, 1, 2, 2, 2, 0, 1, 1, 0, 2]
    x = np.stack([ho.fill_splitmix(n, 90 + i, primes[k]) for i, k in enumerate(which)])
    d = dev(hx, x)
    out = torch.empty_like(d)
    hx.ComputeForwardIndexed(plans, which, out, d, 1, 1)
    got = host(hx, out)
    for i, k in enumerate(which):
        assert (got[i] == ho.NTT(n, primes[k]).forward(x[i], 1, 1)).all(), i
    hx.ComputeInverseIndexed(plans, which, out, out, 1, 1)
    assert (host(hx, out) == x).all()
    with pytest.raises(hx.HexlAmdError):
        hx.ComputeForwardIndexed(plans, [0, 1, 3] + [0] * 8, out, d, 1, 1)  # 3 is not a plan
    with pytest.raises(hx.HexlAmdError):
        hx.ComputeForwardMap(plans, [0, 5], 1, out, d, 1, 1)


_REGISTERED_REGIONS = []


def test_pointer_kinds_and_zero_copy_host_path(hx, ho):
    """hexl_amd_pointer_kind: 0 ordinary host memory, 1 device memory, 2 pinned device-mapped host
    memory (from hexl_amd_host_alloc, or an existing buffer after hexl_amd_host_register); the
    *_host entry points give the oracle's bits on all of them (mapped memory: the kernels run
    straight on the caller's buffer, one-kernel and two-pass transforms, element-wise ops)."""
    import ctypes as C
    lib = hx.lib
    plain = np.zeros(8, dtype=np.uint64)
    assert lib.hexl_amd_pointer_kind(plain.ctypes.data_as(C.c_void_p)) == 0
    assert lib.hexl_amd_pointer_kind(C.c_void_p(dev(hx, plain).data_ptr())) == 1
    for n, bits in ((4096, 49), (8192, 54), (16384, 49), (65536, 54), (131072, 60)):
        q = ho.generate_primes(1, bits, True, n)[0]
        ntt, ont = hx.NTT(n, q), ho.NTT(n, q)
        x = ho.fill_splitmix(n, 5 + n, q)
        want = ont.forward(x, 1, 1)
        pm = C.c_void_p()
        assert lib.hexl_amd_host_alloc(C.byref(pm), 2 * n * 8) == 0
        assert lib.hexl_amd_pointer_kind(pm) == 2
        m = np.ctypeslib.as_array(C.cast(pm, C.POINTER(C.c_uint64)), shape=(2 * n,))
        m[:n] = x
        out = C.c_void_p(pm.value + n * 8)
        assert lib.hexl_amd_pointer_kind(out) == 2  # interior pointers too
        assert lib.hexl_amd_ntt_forward_host(ntt._h, out, pm, 1, 1, 1) == 0
        assert np.array_equal(m[n:], want)
        assert lib.hexl_amd_ntt_inverse_host(ntt._h, out, out, 1, 1, 1) == 0  # in place
        assert np.array_equal(m[n:], x)
        # element-wise on mapped memory: MultMod (op 4)
        assert lib.hexl_amd_eltwise_host(4, out, pm, pm, 0, n, q, 1, 1) == 0
        assert np.array_equal(m[n:], ho.eltwise_mult_mod(x, x, q, 1))
        assert lib.hexl_amd_host_free(pm) == 0
        # an existing allocation, registered: like the memory pool a caller registers once -- its own
        # page-aligned mapping that stays allocated for the life of the process.  (Registering a
        # short-lived heap array and freeing it right after was how this test was written in round
        # 3; the suite then aborted inside the HIP runtime in about one run in twelve, always in the
        # first >= 1 MiB pageable host-to-device copy that followed -- a later allocation reusing
        # the unregistered address range: EXPERIMENTS.md section 9.)
        import mmap
        region = mmap.mmap(-1, 2 * n * 8)
        _REGISTERED_REGIONS.append(region)  # never unmapped
        buf = np.frombuffer(region, dtype=np.uint64)
        buf[:] = 0
        buf[:n] = x
        pb = buf.ctypes.data_as(C.c_void_p)
        assert lib.hexl_amd_host_register(pb, buf.nbytes) == 0
        try:
            assert lib.hexl_amd_pointer_kind(pb) == 2
            po = C.c_void_p(pb.value + n * 8)
            assert lib.hexl_amd_ntt_forward_host(ntt._h, po, pb, 1, 1, 1) == 0
            assert np.array_equal(buf[n:], want)
        finally:
            assert lib.hexl_amd_host_unregister(pb) == 0
        assert lib.hexl_amd_pointer_kind(pb) == 0
    # bounds check entry point on the three kinds
    bad = C.c_uint64(0)
    v = np.array([1, 2, 3, 769, 5, 770], dtype=np.uint64)
    assert lib.hexl_amd_check_bounds(v.ctypes.data_as(C.c_void_p), v.size, 769, C.byref(bad)) == 0
    assert bad.value == 2
    d = dev(hx, v)
    assert lib.hexl_amd_check_bounds(C.c_void_p(d.data_ptr()), v.size, 769, C.byref(bad)) == 0
    assert bad.value == 2
    assert lib.hexl_amd_check_bounds(C.c_void_p(d.data_ptr()), v.size, 771, C.byref(bad)) == 0
    assert bad.value == 0


def test_short_lived_registration_then_large_pageable_copies(hx, ho):
    """The lifetime pattern behind round 4's one-in-eleven abort of this suite, restored as a
    regression test: a short-lived heap array is registered (hexl_amd_host_register), transformed
    in place over the link, unregistered and freed; the next allocations reuse its address range
    and go through >= 1 MiB pageable host-to-device copies (a plan's table upload, a staged *_host
    call, the caller's own copy).  experiments/rocm_fault/register_abort_repro.cpp is the standalone form."""
    import ctypes as C
    lib = hx.lib
    for n, bits in ((4096, 49), (8192, 54), (65536, 54), (65536, 54)):
        q = ho.generate_primes(1, bits, True, n)[0]
        ntt, ont = hx.NTT(n, q), ho.NTT(n, q)
        x = ho.fill_splitmix(n, 5 + n, q)
        want = ont.forward(x, 1, 1)
        buf = np.zeros(2 * n, dtype=np.uint64)  # short-lived; 1 MiB at n = 65536
        buf[:n] = x
        pb = buf.ctypes.data_as(C.c_void_p)
        assert lib.hexl_amd_host_register(pb, buf.nbytes) == 0
        try:
            assert lib.hexl_amd_pointer_kind(pb) == 2
            po = C.c_void_p(pb.value + n * 8)
            assert lib.hexl_amd_ntt_forward_host(ntt._h, po, pb, 1, 1, 1) == 0
            assert np.array_equal(buf[n:], want)
        finally:
            assert lib.hexl_amd_host_unregister(pb) == 0
        assert lib.hexl_amd_pointer_kind(pb) == 0
        del buf, pb, po
    # what followed in the runs that died: large pageable copies from fresh allocations
    n2 = 131072
    q2 = ho.generate_primes(1, 54, True, n2)[0]
    big = hx.NTT(n2, q2)  # 2 x 2 MiB table upload
    y = ho.fill_splitmix(n2, 77, q2)
    src, dst = y.copy(), np.zeros_like(y)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    assert lib.hexl_amd_ntt_forward_host(big._h, p(dst), p(src), 1, 1, 1) == 0  # staged, 1 MiB
    assert np.array_equal(dst, ho.NTT(n2, q2).forward(y, 1, 1))
    z = np.arange(3 << 16, dtype=np.uint64)  # 1.5 MiB through the caller's own copy
    assert np.array_equal(host(hx, dev(hx, z)), z)


@pytest.mark.parametrize("n,batch,bits", [(4096, 1, 49), (4096, 8, 54), (16384, 2, 54), (16384, 3, 54),
                                          (32768, 1, 54), (65536, 1, 54), (1024, 1, 35), (64, 3, 40),
                                          (8192, 1, 54), (8192, 3, 60), (8192, 4, 49), (8192, 5, 28),
                                          (16384, 1, 49), (32768, 2, 60)])
def test_host_pointer_paths_on_ordinary_memory(hx, ho, n, batch, bits):
    """The *_host entry points on ordinary (pageable) host memory, both sides of the 512 KiB
    bounce-buffer threshold -- one-kernel plans in place on the pinned mapped bounce buffer, two-pass plans
    handing over between their passes in device memory (N = 8192 below four polynomials takes the two-pass
    shape there, round 6), staged H2D / D2H above the threshold -- in place and out of place, against the oracle."""
    import ctypes as C
    q = ho.generate_primes(1, bits, True, n)[0]
    ntt, ont = hx.NTT(n, q), ho.NTT(n, q)
    x = np.stack([ho.fill_splitmix(n, 900 + b, q) for b in range(batch)])
    want = ont.forward(x, 1, 1)
    src, dst = x.copy(), np.zeros_like(x)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    assert hx.lib.hexl_amd_ntt_forward_host(ntt._h, p(dst), p(src), batch, 1, 1) == 0
    assert np.array_equal(dst, want) and np.array_equal(src, x)
    assert hx.lib.hexl_amd_ntt_inverse_host(ntt._h, p(dst), p(dst), batch, 1, 1) == 0  # in place
    assert np.array_equal(dst, x)
    # element-wise: MultMod (op 4) and FMAMod with a null addend (op 5), out of place / in place
    a, b = x.reshape(-1).copy(), want.reshape(-1).copy()
    r = np.zeros_like(a)
    assert hx.lib.hexl_amd_eltwise_host(4, p(r), p(a), p(b), 0, a.size, q, 1, 1) == 0
    assert np.array_equal(r, ho.eltwise_mult_mod(a, b, q, 1))
    assert hx.lib.hexl_amd_eltwise_host(5, p(a), p(a), None, 7, a.size, q, 1, 1) == 0
    assert np.array_equal(a, ho.eltwise_fma_mod(x.reshape(-1), 7, None, q, 1))


def test_host_calls_end_on_the_polled_completion_flag(hx, ho):
    """Round 6: a host-pointer call whose last operation is a kernel (the bounce buffer, mapped caller memory)
    learns that it is done from a sequence number a one-thread kernel stores into device-mapped host memory
    ("host_poll", capi.cpp Staging::finish) instead of from hipStreamSynchronize.  Same bits either way and on
    both sides of the bounce limit; the counter tells which wait ended the call."""
    import ctypes as C
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    try:
        for n, bits in ((4096, 49), (16384, 54), (65536, 54)):
            q = ho.generate_primes(1, bits, True, n)[0]
            ntt, ont = hx.NTT(n, q), ho.NTT(n, q)
            x = ho.fill_splitmix(n, 31 + n, q)
            want = ont.forward(x, 1, 1)
            for poll, kb in ((1, 512), (0, 512), (1, 64), (1, 0)):
                hx.set_tuning("host_poll", poll)
                hx.set_tuning("host_bounce_kb", kb)
                bounced = n * 8 <= kb << 10
                before = hx.get_counter("host_polls")
                dst = np.zeros_like(x)
                for _ in range(3):
                    assert hx.lib.hexl_amd_ntt_forward_host(ntt._h, p(dst), p(x), 1, 1, 1) == 0
                    assert np.array_equal(dst, want)
                a = x.copy()
                assert hx.lib.hexl_amd_eltwise_host(4, p(a), p(a), p(want), 0, n, q, 1, 1) == 0
                assert np.array_equal(a, ho.eltwise_mult_mod(x, want, q, 1))
                polled = hx.get_counter("host_polls") - before
                assert polled == (4 if poll and bounced else 0), (n, poll, kb, polled)
        assert hx.get_counter("host_poll_timeouts") == 0
    finally:
        hx.set_tuning("host_poll", 1)
        hx.set_tuning("host_bounce_kb", 512)


@pytest.mark.parametrize("threads", [1, 6])
def test_host_transform_chunk_pipeline(hx, ho, threads):
    """Round 6: a *_host transform of several polynomials and 8 MiB or more of ordinary host memory
    runs as a chunk pipeline (host copy in | H2D + kernels | D2H | host copy out, chunks of whole
    polynomials through four pinned 4 MiB slots, host copies shared by the copy pool): ragged last
    chunk, a polynomial larger than half a slot, in place and out of place, one thread and the
    pool, against the oracle on the first / a middle / the last polynomial and bit for bit against
    the device path; then a small call again (the slots stay big)."""
    import ctypes as C
    import torch
    lib = hx.lib
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    hx.set_tuning("host_copy_threads", threads)
    try:
        for n, batch in ((65536, 21), (16384, 70), (131072 * 4, 5)):  # 10.5 MiB, 8.75 MiB, 20 MiB (one poly = a slot)
            q = ho.generate_primes(1, 54, True, n)[0]
            ntt = hx.NTT(n, q)
            x = np.random.default_rng(n + batch).integers(0, q, (batch, n), dtype=np.uint64)
            d = hx.from_numpy(x)
            ntt.ComputeForward(d, d, 1, 1)
            want = hx.to_numpy(d)
            rows = sorted({0, batch // 2, batch - 1})
            assert np.array_equal(want[rows], ho.NTT(n, q).forward(x[rows], 1, 1))
            src, dst = x.copy(), np.zeros_like(x)
            assert lib.hexl_amd_ntt_forward_host(ntt._h, p(dst), p(src), batch, 1, 1) == 0
            assert np.array_equal(dst, want) and np.array_equal(src, x)
            assert lib.hexl_amd_ntt_inverse_host(ntt._h, p(dst), p(dst), batch, 1, 1) == 0  # in place
            assert np.array_equal(dst, x)
            del d
            torch.cuda.empty_cache()
        n = 4096
        q = ho.generate_primes(1, 49, True, n)[0]
        ntt = hx.NTT(n, q)
        x = ho.fill_splitmix(n, 5, q)
        y = x.copy()
        assert lib.hexl_amd_ntt_forward_host(ntt._h, p(y), p(y), 1, 1, 1) == 0
        assert np.array_equal(y, ho.NTT(n, q).forward(x, 1, 1))
    finally:
        hx.set_tuning("host_copy_threads", 6)


def test_host_transforms_from_several_threads(hx, ho):
    """Four host threads calling the *_host transform at once on their own 12 MiB buffers of ordinary memory: each has
    its own pinned slots, streams and chunk pipeline; the copy pool serves one of them at a time (the others copy by
    themselves).  Every result equals the device path's."""
    import ctypes as C
    import threading
    n, batch = 32768, 48
    q = ho.generate_primes(1, 54, True, n)[0]
    ntt = hx.NTT(n, q)
    xs = [np.random.default_rng(100 + t).integers(0, q, (batch, n), dtype=np.uint64) for t in range(4)]
    want = []
    for x in xs:
        d = hx.from_numpy(x)
        ntt.ComputeForward(d, d, 1, 1)
        want.append(hx.to_numpy(d))
    assert np.array_equal(want[0][[0, batch - 1]], ho.NTT(n, q).forward(xs[0][[0, batch - 1]], 1, 1))
    outs = [np.zeros_like(x) for x in xs]
    errors = []

    def work(t):
        for _ in range(3):
            rc = hx.lib.hexl_amd_ntt_forward_host(ntt._h, outs[t].ctypes.data_as(C.c_void_p),
                                                  xs[t].ctypes.data_as(C.c_void_p), batch, 1, 1)
            if rc != 0 or not np.array_equal(outs[t], want[t]):
                errors.append((t, rc))
    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


@pytest.mark.parametrize("direct", [0, 1])
def test_host_pointer_paths_above_one_mebibyte(hx, ho, direct):
    """Host-pointer calls whose buffers exceed one pinned slot (1 MiB): by default the library
    copies ordinary caller memory through its own pinned slots, chunk by chunk, in both
    directions ("host_direct_copy" 0: the HIP runtime never gets to pin the caller's pages, the
    path the suite's GPU memory access faults sat in); with the key at 1 the buffers go to
    hipMemcpyAsync whole.  Ragged sizes (not a multiple of the slot), in place and out of place,
    NTT / element-wise / DyadicMultiply / hexl_amd_copy, against the oracle."""
    import ctypes as C
    lib = hx.lib
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    hx.set_tuning("host_direct_copy", direct)
    try:
        for n, batch in ((131072, 1), (65536, 5), (4096, 200)):  # 1 MiB, 2.5 MiB, 6.25 MiB
            q = ho.generate_primes(1, 54, True, n)[0]
            ntt, ont = hx.NTT(n, q), ho.NTT(n, q)
            x = np.stack([ho.fill_splitmix(n, 40 + b, q) for b in range(batch)])
            rows = sorted({0, batch // 2, batch - 1})
            want = ont.forward(x[rows], 1, 1)
            src, dst = x.copy(), np.zeros_like(x)
            assert lib.hexl_amd_ntt_forward_host(ntt._h, p(dst), p(src), batch, 1, 1) == 0
            assert np.array_equal(dst[rows], want) and np.array_equal(src, x)
            assert lib.hexl_amd_ntt_inverse_host(ntt._h, p(dst), p(dst), batch, 1, 1) == 0  # in place
            assert np.array_equal(dst, x)
        # element-wise MultMod over 3 MiB + 8 bytes per operand
        q = ho.generate_primes(1, 58, True, 2)[0]
        m = (3 << 17) + 1
        a, b = ho.fill_splitmix(m, 1, q), ho.fill_splitmix(m, 2, q)
        r = np.zeros_like(a)
        assert lib.hexl_amd_eltwise_host(4, p(r), p(a), p(b), 0, m, q, 1, 1) == 0
        assert np.array_equal(r, ho.eltwise_mult_mod(a, b, q, 1))
        # DyadicMultiply: operands 2 x (n x k) words = 1.5 MiB each, result 2.25 MiB, in place on x
        n, k = 32768, 3
        moduli = [int(v) for v in ho.generate_primes(k, 50, True, n)]
        xs = np.concatenate([ho.fill_splitmix(n, 10 + i, moduli[i % k]) for i in range(2 * k)])
        ys = np.concatenate([ho.fill_splitmix(n, 20 + i, moduli[i % k]) for i in range(2 * k)])
        want = ho.dyadic_multiply(xs, ys, n, moduli)
        out = np.zeros(3 * n * k, dtype=np.uint64)
        mod = (C.c_uint64 * k)(*moduli)
        assert lib.hexl_amd_dyadic_multiply_host(p(out), p(xs), p(ys), n, mod, k) == 0
        assert np.array_equal(out, want)
        # hexl_amd_copy: pageable -> device -> pageable, 2.5 MiB + 8 bytes, non-blocking calls
        z = np.arange((5 << 16) + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        back = np.zeros_like(z)
        d = C.c_void_p()
        assert lib.hexl_amd_device_alloc(C.byref(d), z.nbytes, -1) == 0
        st = C.c_void_p()
        assert lib.hexl_amd_stream_create(C.byref(st), -1) == 0
        assert lib.hexl_amd_copy(d, p(z), z.nbytes, st, 0) == 0
        assert lib.hexl_amd_copy(p(back), d, z.nbytes, st, 0) == 0
        assert lib.hexl_amd_synchronize(st) == 0
        assert np.array_equal(back, z)
        assert lib.hexl_amd_stream_destroy(st) == 0 and lib.hexl_amd_device_free(d) == 0
    finally:
        hx.set_tuning("host_direct_copy", 0)


def test_synchronize_of_a_library_stream_polls_a_completion_flag(hx, ho):
    """hexl_amd_synchronize on a stream from hexl_amd_stream_create learns that the stream is done from a sequence
    number a one-thread kernel publishes in mapped host memory ("host_poll"; round 6) -- the results are there when it
    returns, from several threads on ONE stream too, and with the key off it waits in the runtime as before."""
    import ctypes as C
    import threading
    lib = hx.lib
    n, q = 4096, int(ho.generate_primes(1, 49, True, 4096)[0])
    ntt, ont = hx.NTT(n, q), ho.NTT(n, q)
    st = C.c_void_p()
    assert lib.hexl_amd_stream_create(C.byref(st), -1) == 0
    x = ho.fill_splitmix(n, 3, q)
    want = ont.forward(x, 1, 1)
    bufs = []
    for _ in range(4):
        d = C.c_void_p()
        assert lib.hexl_amd_device_alloc(C.byref(d), n * 8, -1) == 0
        bufs.append(d)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    try:
        for poll in (1, 0):
            hx.set_tuning("host_poll", poll)
            before = hx.get_counter("host_polls")
            for _ in range(20):
                assert lib.hexl_amd_copy(bufs[0], p(x), n * 8, st, 0) == 0
                assert lib.hexl_amd_ntt_forward(ntt._h, bufs[0], bufs[0], 1, 1, 1, st) == 0
                back = np.zeros_like(x)
                assert lib.hexl_amd_copy(p(back), bufs[0], n * 8, st, 0) == 0
                assert lib.hexl_amd_synchronize(st) == 0
                assert np.array_equal(back, want)
            assert (hx.get_counter("host_polls") - before >= 20) == bool(poll)
        hx.set_tuning("host_poll", 1)
        errors = []

        def worker(t):
            mine = np.zeros_like(x)
            for k in range(100):
                if lib.hexl_amd_copy(bufs[t], p(x), n * 8, st, 0) or \
                        lib.hexl_amd_ntt_forward(ntt._h, bufs[t], bufs[t], 1, 1, 1, st) or \
                        lib.hexl_amd_copy(p(mine), bufs[t], n * 8, st, 0) or lib.hexl_amd_synchronize(st):
                    errors.append((t, k, "rc"))
                    return
                if not np.array_equal(mine, want):
                    errors.append((t, k, "result"))
                    return
        ts = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors[:3]
    finally:
        hx.set_tuning("host_poll", 1)
        assert lib.hexl_amd_stream_destroy(st) == 0
        for d in bufs:
            assert lib.hexl_amd_device_free(d) == 0


def test_ntt_config1_on_the_hip_path(hx, ho):
    """BASELINE configs[0] at its exact parameters on the GPU: N = 1024, q = 0xffffee001
    (36-bit), ONE polynomial, seed 1, Fwd(1,1) + Inv(1,1) against the oracle; the plan picks
    the minimal root the survey recorded from the real reference (46310425)."""
    n, q = 1024, 0xffffee001
    ntt, ont = hx.NTT(n, q), ho.NTT(n, q)
    assert ntt.GetMinimalRootOfUnity() == 46310425
    x = ho.fill_splitmix(n, 1, q)
    d = dev(hx, x)
    ntt.ComputeForward(d, d, 1, 1)
    f = host(hx, d)
    assert (f == ont.forward(x, 1, 1)).all()
    assert (f < np.uint64(q)).all()
    ntt.ComputeInverse(d, d, 1, 1)
    assert (host(hx, d) == x).all()
    # out of place, lazy ranges
    import torch
    d = dev(hx, x)
    out = torch.empty_like(d)
    ntt.ComputeForward(out, d, 1, 4)
    lz = host(hx, out)
    assert (lz < np.uint64(4 * q)).all() and (lz % np.uint64(q) == f).all()
    ntt.ComputeInverse(out, dev(hx, f), 1, 2)
    lz = host(hx, out)
    assert (lz < np.uint64(2 * q)).all() and (lz % np.uint64(q) == x).all()


DEFN = json.load(open(os.path.join(os.path.dirname(__file__), "golden",
                                   "ntt_definition_fixtures.json")))


@pytest.mark.parametrize("case", DEFN["cases"], ids=lambda c: "n%d" % c["n"])
def test_ntt_matches_definition_fixtures(hx, case):
    """HIP output against known answers computed from the transform's definition in big
    integers (tests/golden/make_ntt_definition_fixtures.py) at N = 4096 / 65536 / 131072 --
    directly, not through the oracle: 64 sampled entries + the digest of the whole vector,
    forward and inverse; the polynomial is the second of a batch of 5."""
    import hashlib
    import torch
    n, q = case["n"], case["q"]
    ntt = hx.NTT(n, q)
    assert ntt.GetMinimalRootOfUnity() == case["minimal_root"]
    for name, fn in (("forward", ntt.ComputeForward), ("inverse", ntt.ComputeInverse)):
        seed = case[name]["seed"]
        x = torch.empty((5, n), dtype=torch.int64, device="cuda")
        hx.fill_splitmix(x, n, 5, seed - 1, q)  # polynomial 1 of the batch = splitmix(seed)
        fn(x, x, 1, 1)
        got = host(hx, x[1])
        for i, v in case[name]["samples"]:
            assert int(got[i]) == v
        assert hashlib.sha256(got.astype("<u8").tobytes()).hexdigest() == case[name]["sha256_le_u64"]


def test_ntt_config2_full_batch(hx, ho):
    """BASELINE configs[1]: N=4096, the survey's 50-bit prime, batch 256 -- every
    polynomial against the oracle, forward and inverse."""
    import torch
    n, batch = 4096, 256
    q = KAT["generate_primes_survey_probe"]["cases"][0]["out"][0]
    assert q == 562949954093057
    x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
    hx.fill_splitmix(x, n, batch, 31, q)
    hxin = host(hx, x)
    gnt, ont = hx.NTT(n, q), ho.NTT(n, q)
    y = torch.empty_like(x)
    gnt.ComputeForward(y, x, 1, 1)
    ref = ont.forward(hxin, 1, 1)
    assert (host(hx, y) == ref).all()
    gnt.ComputeInverse(y, y, 1, 1)
    assert torch.equal(x, y)
    z = torch.empty_like(x)
    gnt.ComputeInverse(z, x, 1, 1)
    assert (host(hx, z) == ont.inverse(hxin, 1, 1)).all()


def test_ntt_config4_full_size_one_gpu(hx, ho):
    """BASELINE configs[3] on one GPU: 8 RNS primes x 4096 polynomials of N=65536
    (16 GiB) through the RNS entry point; oracle spot checks per prime, value range,
    round trip against regenerated input."""
    import torch
    n, B = 65536, 4096
    primes = KAT["generate_primes_survey_probe"]["cases"][1]["out"]
    plans = [hx.NTT(n, p) for p in primes]
    x = torch.empty((len(primes), B, n), dtype=torch.int64, device="cuda")
    for k, p in enumerate(primes):
        hx.fill_splitmix(x[k], n, B, 1 + k * B, p)
    hx.ComputeForwardRNS(plans, x, x, 1, 1)
    for k, p in enumerate(primes):
        ont = ho.NTT(n, p)
        assert int(x[k].min()) >= 0 and int(x[k].max()) < p
        for b in (0, 1 + 511 * k, B - 1):
            assert (host(hx, x[k, b]) == ont.forward(ho.fill_splitmix(n, 1 + k * B + b, p), 1, 1)).all()
    hx.ComputeInverseRNS(plans, x, x, 1, 1)
    chunk = torch.empty((B, n), dtype=torch.int64, device="cuda")
    for k, p in enumerate(primes):
        hx.fill_splitmix(chunk, n, B, 1 + k * B, p)
        assert torch.equal(chunk, x[k])


@pytest.mark.parametrize("n,bits,small_end", [(1 << 20, 55, False), (1 << 20, 29, False),
                                             (1 << 16, 29, False), (1 << 16, 30, True),
                                             (1 << 16, 49, False), (1 << 20, 49, False),
                                             (1 << 12, 49, False), (1 << 16, 50, True),
                                             (1 << 16, 55, False), (1 << 16, 56, True),
                                             (1 << 16, 59, False), (1 << 20, 59, False),
                                             (1 << 12, 59, False), (1 << 16, 60, True),
                                             (1 << 17, 60, True), (1 << 16, 60, False),
                                             (1 << 17, 61, False), (1 << 20, 61, False)])
def test_ntt_policy_boundaries(hx, ho, n, bits, small_end):
    """Moduli at the edges of the five arithmetic policies: just below 2^30 (Small) and
    just above it (Fp64), just below 2^50 (Fp64: exact integers in doubles, 7-stage
    forward runs reach 7.9 q < 2^53) and just above it (Lazy), just below 2^56 (Lazy:
    with input_mod_factor 4 at N = 2^20 the doubled values reach (8 + 6*20) q = 128 q,
    just under 2^63) and just above it (Harvey60), just below 2^60 and just above it
    (Harvey60 up to 2^60 + 2^28: doubled values up to 8 q <= 2^63 + 2^31, the high word
    at most 2^31), just below 2^61 and just below 2^62 (Strict; the largest moduli the
    API admits).  All (in, out) factors."""
    q = ho.generate_primes(1, bits, small_end, n)[0]
    lo, hi = 1 << bits, 1 << (bits + 1)
    assert lo < q < hi and ((q - lo) < (hi - lo) // 8 if small_end else (hi - q) < (hi - lo) // 8)
    ont, gnt = ho.NTT(n, q), hx.NTT(n, q)
    for in_mf, out_mf in ((4, 4), (4, 1), (1, 1)):
        x = ho.fill_splitmix(n, 5 + in_mf, in_mf * q)
        # the largest legal inputs as well
        x[:4] = np.uint64(in_mf * q - 1)
        ref = ont.forward(x % np.uint64(q), 1, 1)
        d = dev(hx, x)
        gnt.ComputeForward(d, d, in_mf, out_mf)
        got = host(hx, d)
        assert (got < out_mf * q).all() and ((got % np.uint64(q)) == ref).all()
        if out_mf == 1:
            assert (got == ref).all()
    for in_mf, out_mf in ((2, 2), (2, 1), (1, 1)):
        x = ho.fill_splitmix(n, 9 + in_mf, in_mf * q)
        x[:4] = np.uint64(in_mf * q - 1)
        ref = ont.inverse(x % np.uint64(q), 1, 1)
        d = dev(hx, x)
        gnt.ComputeInverse(d, d, in_mf, out_mf)
        got = host(hx, d)
        assert (got < out_mf * q).all() and ((got % np.uint64(q)) == ref).all()
        if out_mf == 1:
            assert (got == ref).all()


@pytest.mark.parametrize("n,batch", [(4096, 256), (65536, 64), (1 << 17, 3), (1 << 13, 5)])
def test_ntt_fp64_policy_matches_integer_policy(hx, n, batch):
    """q < 2^50 (the reference's IFMA / FP64-class moduli): the Fp64 arithmetic policy
    (doubles, balanced twiddles) against the integer Lazy policy on the same inputs, bit
    for bit -- plans built with the policy switched on and off."""
    import torch
    q = 562949954093057 if n == 4096 else hx.GeneratePrimes(1, 49, False, n)[0]
    try:
        hx.set_tuning("fp64", 0)
        lazy = hx.NTT(n, q)
        hx.set_tuning("fp64", 1)
        fp = hx.NTT(n, q)
    finally:
        hx.set_tuning("fp64", 1)
    for in_mf in (1, 4):
        x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
        hx.fill_splitmix(x, n, batch, 77, in_mf * q)
        a, b = torch.empty_like(x), torch.empty_like(x)
        lazy.ComputeForward(a, x, in_mf, 1)
        fp.ComputeForward(b, x, in_mf, 1)
        assert torch.equal(a, b)
        fp.ComputeForward(b, x, in_mf, 4)
        assert int(b.min()) >= 0 and int(b.max()) < 4 * q and torch.equal(b % q, a)
    for in_mf in (1, 2):
        hx.fill_splitmix(x, n, batch, 78, in_mf * q)
        lazy.ComputeInverse(a, x, in_mf, 1)
        fp.ComputeInverse(b, x, in_mf, 1)
        assert torch.equal(a, b)
        fp.ComputeInverse(x, x, in_mf, 2)  # in place, lazy output range
        assert int(x.min()) >= 0 and int(x.max()) < 2 * q and torch.equal(x % q, a)


@pytest.mark.parametrize("n,batch,bits,small_end", [
    (64, 9, 30, True), (1024, 33, 35, True), (4096, 256, 36, True), (4096, 64, 46, False),
    (8192, 40, 43, True), (8192, 7, 46, False), (16384, 200, 44, True), (16384, 5, 46, False),
    (32768, 12, 40, True), (65536, 64, 46, False), (65536, 20, 33, True), (1 << 17, 3, 46, False),
    (1 << 18, 2, 45, True), (1 << 19, 2, 46, False), (1 << 20, 1, 46, False)])
def test_ntt_fp64_long_run_policy(hx, n, batch, bits, small_end):
    """2^30 <= q < 2^47 (SEAL's default moduli up to N = 8192): the long-run member of the Fp64
    family (modarith.h Fp64L: no reduction inside a forward pass, inverse runs of 6 stages instead
    of 3) against the integer policy AND against the short-run Fp64 on the same inputs, bit for
    bit -- primes at both ends of the range, adversarial inputs, every plan shape."""
    import torch
    q = hx.GeneratePrimes(1, bits, small_end, n)[0]
    assert (1 << 30) <= q < (1 << 47)
    try:
        hx.set_tuning("fp64", 0)
        integer = hx.NTT(n, q)
        hx.set_tuning("fp64", 1)
        hx.set_tuning("fp64_long", 0)
        short = hx.NTT(n, q)
        hx.set_tuning("fp64_long", 1)
        long_run = hx.NTT(n, q)
    finally:
        hx.set_tuning("fp64", 1)
        hx.set_tuning("fp64_long", 1)
    x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
    a, b, c = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    for in_mf in (1, 2, 4):
        for fill in ("random", "top"):
            if fill == "random":
                hx.fill_splitmix(x, n, batch, 171, in_mf * q)
            else:
                x.fill_(in_mf * q - 1)
            integer.ComputeForward(a, x, in_mf, 1)
            short.ComputeForward(b, x, in_mf, 1)
            long_run.ComputeForward(c, x, in_mf, 1)
            assert torch.equal(a, b) and torch.equal(a, c)
            long_run.ComputeForward(c, x, in_mf, 4)
            assert int(c.min()) >= 0 and int(c.max()) < 4 * q and torch.equal(c % q, a)
    for in_mf in (1, 2):
        for fill in ("random", "top"):
            if fill == "random":
                hx.fill_splitmix(x, n, batch, 172, in_mf * q)
            else:
                x.fill_(in_mf * q - 1)
            integer.ComputeInverse(a, x, in_mf, 1)
            short.ComputeInverse(b, x, in_mf, 1)
            long_run.ComputeInverse(c, x, in_mf, 1)
            assert torch.equal(a, b) and torch.equal(a, c)
            long_run.ComputeInverse(c, x, in_mf, 2)
            assert int(c.min()) >= 0 and int(c.max()) < 2 * q and torch.equal(c % q, a)


@pytest.mark.parametrize("n,bits,small_end", [(4096, 30, True), (65536, 31, False), (16384, 31, True),
                                              (8192, 32, True), (65536, 32, True)])
def test_ntt_integer_policies_around_2_pow_32(hx, ho, n, bits, small_end):
    """With the Fp64 policy switched off the moduli between 2^30 and 2^32 take the Harvey60
    policy and the Lazy policy starts at 2^32 (its quotient estimates shift the high word of a
    value: round 3); both against the oracle and against the Fp64 plan, all mod factors."""
    import torch
    q = ho.generate_primes(1, bits, small_end, n)[0]
    try:
        hx.set_tuning("fp64", 0)
        integer = hx.NTT(n, q)
    finally:
        hx.set_tuning("fp64", 1)
    fp = hx.NTT(n, q)
    ont = ho.NTT(n, q)
    batch = 3
    for in_mf, out_mf in ((1, 1), (4, 1), (2, 4)):
        x = np.stack([ho.fill_splitmix(n, 300 + b, in_mf * q) for b in range(batch)])
        want = ont.forward(x, in_mf, 1)
        a, b = dev(hx, x), dev(hx, x)
        integer.ComputeForward(a, a, in_mf, out_mf)
        fp.ComputeForward(b, b, in_mf, out_mf)
        for got in (host(hx, a), host(hx, b)):
            assert (got < np.uint64(out_mf * q)).all() and (got % np.uint64(q) == want).all()
            if out_mf == 1:
                assert (got == want).all()
    for in_mf, out_mf in ((1, 1), (2, 1), (2, 2)):
        x = np.stack([ho.fill_splitmix(n, 400 + b, in_mf * q) for b in range(batch)])
        want = ont.inverse(x, in_mf, 1)
        a = dev(hx, x)
        integer.ComputeInverse(a, a, in_mf, out_mf)
        got = host(hx, a)
        assert (got < np.uint64(out_mf * q)).all() and (got % np.uint64(q) == want).all()


@pytest.mark.parametrize("n,batch,bits,small_end", [(4096, 64, 59, False), (65536, 64, 59, False),
                                                    (1 << 17, 3, 56, True), (1 << 13, 5, 58, True),
                                                    (1 << 14, 200, 59, False), (64, 7, 57, True),
                                                    (1 << 17, 8, 60, True), (4096, 33, 60, True)])
def test_ntt_harvey60_policy_matches_strict_policy(hx, n, batch, bits, small_end):
    """2^56 <= q < 2^60 + 2^28 (SEAL's and OpenFHE's 60-bit primes, and the smallest primes
    above 2^60 the reference's own tests generate): the Harvey60 arithmetic policy
    (Harvey ranges on doubled values, carry-free products) against the Strict policy on the
    same inputs, bit for bit -- plans built with the policy switched on and off."""
    import torch
    q = hx.GeneratePrimes(1, bits, small_end, n)[0]
    try:
        hx.set_tuning("lazy_family", 0)  # (below 2^59 the bounded Lazy members come first)
        hx.set_tuning("h60", 0)
        strict = hx.NTT(n, q)
        hx.set_tuning("h60", 1)
        h60 = hx.NTT(n, q)
    finally:
        hx.set_tuning("h60", 1)
        hx.set_tuning("lazy_family", 1)
    x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
    for in_mf in (1, 2, 4):
        hx.fill_splitmix(x, n, batch, 81, in_mf * q)
        a, b = torch.empty_like(x), torch.empty_like(x)
        strict.ComputeForward(a, x, in_mf, 1)
        h60.ComputeForward(b, x, in_mf, 1)
        assert torch.equal(a, b)
        h60.ComputeForward(b, x, in_mf, 4)
        assert int(b.min()) >= 0 and int(b.max()) < 4 * q and torch.equal(b % q, a)
    for in_mf in (1, 2):
        hx.fill_splitmix(x, n, batch, 82, in_mf * q)
        strict.ComputeInverse(a, x, in_mf, 1)
        h60.ComputeInverse(b, x, in_mf, 1)
        assert torch.equal(a, b)
        h60.ComputeInverse(x, x, in_mf, 2)  # in place, lazy output range
        assert int(x.min()) >= 0 and int(x.max()) < 2 * q and torch.equal(x % q, a)


@pytest.mark.parametrize("n,batch,bits,small_end", [
    (64, 7, 56, True), (4096, 64, 56, True), (4096, 33, 57, False), (1 << 13, 5, 56, False),
    (1 << 14, 200, 57, True), (1 << 14, 3, 57, False), (65536, 64, 56, True), (65536, 16, 57, False),
    (1 << 17, 8, 57, True), (1 << 20, 1, 57, False), (1 << 18, 2, 56, True),           # Lazy32: [2^56, 2^58)
    (64, 7, 58, True), (4096, 64, 58, True), (4096, 33, 58, False), (1 << 13, 5, 58, False),
    (1 << 14, 200, 58, True), (65536, 64, 58, True), (65536, 16, 58, False), (1 << 17, 8, 58, True),
    (1 << 20, 1, 58, False), (1 << 19, 2, 58, True)])                                  # Lazy16: [2^58, 2^59)
def test_ntt_bounded_lazy_policies_match_strict_policy(hx, n, batch, bits, small_end):
    """2^56 <= q < 2^59: the bounded members of the Lazy family (modarith.h LazyT: doubled values
    that stay below 32q / 16q -- the forward network subtracts 16q / 8q from the x operands of
    the stages the host marks, the inverse network runs lazy_inverse.h's schedule for the smaller
    limit) against the Strict policy on the same inputs, bit for bit, at both ends of each range,
    one-kernel, two-pass and three-pass plans, every legal factor pair."""
    import torch
    q = hx.GeneratePrimes(1, bits, small_end, n)[0]
    assert (1 << 56) <= q < (1 << 59)
    try:
        hx.set_tuning("lazy_family", 0)
        hx.set_tuning("h60", 0)
        strict = hx.NTT(n, q)
    finally:
        hx.set_tuning("h60", 1)
        hx.set_tuning("lazy_family", 1)
    lazy = hx.NTT(n, q)
    x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
    for in_mf in (1, 2, 4):
        hx.fill_splitmix(x, n, batch, 91, in_mf * q)
        a, b = torch.empty_like(x), torch.empty_like(x)
        strict.ComputeForward(a, x, in_mf, 1)
        lazy.ComputeForward(b, x, in_mf, 1)
        assert torch.equal(a, b)
        lazy.ComputeForward(b, x, in_mf, 4)
        assert int(b.min()) >= 0 and int(b.max()) < 4 * q and torch.equal(b % q, a)
    for in_mf in (1, 2):
        hx.fill_splitmix(x, n, batch, 92, in_mf * q)
        strict.ComputeInverse(a, x, in_mf, 1)
        lazy.ComputeInverse(b, x, in_mf, 1)
        assert torch.equal(a, b)
        lazy.ComputeInverse(x, x, in_mf, 2)  # in place, lazy output range
        assert int(x.min()) >= 0 and int(x.max()) < 2 * q and torch.equal(x % q, a)
    # adversarial inputs: every coefficient at the top of its range
    for in_mf in (1, 4):
        x.fill_(in_mf * q - 1)
        strict.ComputeForward(a, x, in_mf, 1)
        lazy.ComputeForward(b, x, in_mf, 1)
        assert torch.equal(a, b)
    x.fill_(2 * q - 1)
    strict.ComputeInverse(a, x, 2, 1)
    lazy.ComputeInverse(b, x, 2, 1)
    assert torch.equal(a, b)


@pytest.mark.parametrize("logn", [13, 14])
@pytest.mark.parametrize("bits", [28, 45, 54, 60])
def test_ntt_single_kernel_plans(hx, ho, logn, bits):
    """N = 8192 / 16384 as ONE kernel on a 64 / 128 KiB LDS tile (16 elements per thread and
    rounds of four stages at N = 16384; batches >= 96 there) against the two-pass plan, bit
    for bit, and against the oracle; every arithmetic policy, lazy outputs, out of place."""
    import torch
    n, batch = 1 << logn, 200
    q = ho.generate_primes(1, bits, True, n)[0]
    gnt, ont = hx.NTT(n, q), ho.NTT(n, q)
    x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
    try:
        for fwd in (True, False):
            fn = gnt.ComputeForward if fwd else gnt.ComputeInverse
            for in_mf, out_mf in (((1, 1), (4, 4)) if fwd else ((1, 1), (2, 2))):
                hx.fill_splitmix(x, n, batch, 5 + in_mf, in_mf * q)
                res = []
                for t13 in (0, 2):
                    hx.set_tuning("tile13", t13)
                    a = x.clone()
                    fn(a, a, in_mf, out_mf)
                    b = torch.full_like(x, -1)
                    fn(b, x, in_mf, out_mf)
                    assert torch.equal(a, b)
                    res.append(a)
                if out_mf == 1:
                    assert torch.equal(res[0], res[1])
                    ref = (ont.forward if fwd else ont.inverse)(host(hx, x[[0, 77, 199]]) % np.uint64(q), 1, 1)
                    assert (host(hx, res[1][[0, 77, 199]]) == ref).all()
                else:
                    assert torch.equal(res[0] % q, res[1] % q)
                    assert int(res[1].min()) >= 0 and int(res[1].max()) < out_mf * q
    finally:
        hx.set_tuning("tile13", 2)


@pytest.mark.parametrize("bits", [28, 44, 49, 54, 56, 58, 60, 61])
def test_ntt_tile_walk_n16384(hx, ho, bits):
    """Round 6: the N = 16384 one-kernel plan as a persistent workgroup per compute unit that walks
    the polynomials (tile_walk, `walk14`), forced on for every arithmetic policy and both directions:
    a batch of more polynomials than compute units that is no multiple of the grid, in place and out
    of place, canonical outputs against the oracle (first, a middle and the last polynomial of the
    walk) and bit for bit against one workgroup per polynomial; lazy outputs congruent and in range."""
    import torch
    n, batch = 16384, 601
    q = ho.generate_primes(1, bits, True, n)[0]
    gnt, ont = hx.NTT(n, q), ho.NTT(n, q)
    x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
    probe = [0, 255, 256, 300, 600]
    try:
        for fwd in (True, False):
            fn = gnt.ComputeForward if fwd else gnt.ComputeInverse
            for in_mf, out_mf in (((1, 1), (4, 4)) if fwd else ((1, 1), (2, 2))):
                hx.fill_splitmix(x, n, batch, 9 + in_mf, in_mf * q)
                res = []
                for walk in (0, 2):
                    hx.set_tuning("walk14", walk)
                    a = x.clone()
                    fn(a, a, in_mf, out_mf)
                    b = torch.full_like(x, -1)
                    fn(b, x, in_mf, out_mf)
                    assert torch.equal(a, b)
                    res.append(a)
                if out_mf == 1:
                    assert torch.equal(res[0], res[1])
                    ref = (ont.forward if fwd else ont.inverse)(host(hx, x[probe]) % np.uint64(q), 1, 1)
                    assert (host(hx, res[1][probe]) == ref).all()
                else:
                    assert torch.equal(res[0] % q, res[1] % q)
                    assert int(res[1].min()) >= 0 and int(res[1].max()) < out_mf * q
    finally:
        hx.set_tuning("walk14", 1)


def test_ntt_tile_walk_rns_and_default_table(hx, ho):
    """The multi-plan form of the walk (several moduli in one launch: RNS limbs of different
    arithmetic policies) against per-plan calls, and the default `walk14` table (inverse: every
    policy; forward: the Fp64 policies) against the oracle on a batch the walk serves."""
    import torch
    n, per = 16384, 150
    moduli = [ho.generate_primes(1, b, True, n)[0] for b in (44, 54, 54 + 1, 60)]
    moduli[2] = ho.generate_primes(2, 54, True, n)[1]
    plans = [hx.NTT(n, q) for q in moduli]
    x = torch.empty((len(moduli) * per, n), dtype=torch.int64, device="cuda")
    for k, q in enumerate(moduli):
        hx.fill_splitmix(x[k * per:(k + 1) * per], n, per, 21 + k, q)
    try:
        out = {}
        for walk in (0, 1, 2):
            hx.set_tuning("walk14", walk)
            f = torch.empty_like(x)
            hx.ComputeForwardRNS(plans, f, x, 1, 1)
            i = torch.empty_like(x)
            hx.ComputeInverseRNS(plans, i, f, 1, 1)
            assert torch.equal(i, x)
            out[walk] = f
        assert torch.equal(out[0], out[1]) and torch.equal(out[0], out[2])
        for k, q in enumerate(moduli):
            rows = [k * per, k * per + per - 1]
            assert (host(hx, out[1][rows]) == ho.NTT(n, q).forward(host(hx, x[rows]), 1, 1)).all()
    finally:
        hx.set_tuning("walk14", 1)


def test_ntt_headline_full_size_properties(hx, ho):
    """BASELINE configs[2] at full size: N=65536, 55-bit q, batch=4096 (2 GiB).
    Size-independent properties on the device plus oracle spot checks."""
    import torch
    n, batch = 65536, 4096
    q = KAT["generate_primes_survey_probe"]["cases"][1]["out"][0]
    ntt = hx.NTT(n, q)
    x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
    hx.fill_splitmix(x, n, batch, 1, q)
    # device generator == oracle generator
    assert (host(hx, x[17]) == ho.fill_splitmix(n, 1 + 17, q)).all()
    y = torch.empty_like(x)
    ntt.ComputeForward(y, x, 1, 1)
    assert int(y.min()) >= 0 and int(y.max()) < q
    ont = ho.NTT(n, q)
    for b in (0, 1, 2047, 4095):
        assert (host(hx, y[b]) == ont.forward(host(hx, x[b]), 1, 1)).all()
    # linearity: NTT(x_b + x_{b+1}) == NTT(x_b) + NTT(x_{b+1}) on 64 pairs
    s = torch.empty((64, n), dtype=torch.int64, device="cuda")
    hx.EltwiseAddMod(s, x[:64].contiguous(), x[64:128].contiguous(), 64 * n, q)
    fs = torch.empty_like(s)
    ntt.ComputeForward(fs, s, 1, 1)
    fsum = torch.empty_like(s)
    hx.EltwiseAddMod(fsum, y[:64].contiguous(), y[64:128].contiguous(), 64 * n, q)
    assert torch.equal(fs, fsum)
    # round trip in place over the whole batch
    ntt.ComputeInverse(y, y, 1, 1)
    assert torch.equal(x, y)


# ---------------------------------------------------------------- eltwise KATs
@pytest.mark.parametrize("case", KAT["eltwise_mult_mod"]["cases"])
def test_mult_mod_kat(hx, ho, case):
    import torch
    q = resolve_q(ho, case["q"])
    a, b, exp = (resolve(case[k], q) for k in ("a", "b", "out"))
    da, db = dev(hx, a), dev(hx, b)
    out = torch.zeros_like(da)
    hx.EltwiseMultMod(out, da, db, len(a), q, case["in_mf"])
    assert host(hx, out).tolist() == exp
    hx.EltwiseMultMod(da, da, db, len(a), q, case["in_mf"])  # in place
    assert host(hx, da).tolist() == exp


@pytest.mark.parametrize("case", KAT["eltwise_fma_mod"]["cases"])
def test_fma_mod_kat(hx, ho, case):
    q = resolve_q(ho, case["q"])
    da = dev(hx, case["a"])
    dc = dev(hx, case["c"]) if case["c"] is not None else None
    hx.EltwiseFMAMod(da, da, case["s"], dc, len(case["a"]), q, case["in_mf"])
    assert host(hx, da).tolist() == case["out"]


@pytest.mark.parametrize("case", KAT["eltwise_reduce_mod"]["cases"])
def test_reduce_mod_kat(hx, case):
    import torch
    q = case["q"]
    in_mf = q if case["in_mf"] == "q" else case["in_mf"]
    da = dev(hx, case["a"])
    out = torch.zeros_like(da)
    hx.EltwiseReduceMod(out, da, len(case["a"]), q, in_mf, case["out_mf"])
    assert host(hx, out).tolist() == case["out"]


@pytest.mark.parametrize("case", KAT["eltwise_add_mod"]["cases"])
def test_add_mod_kat(hx, ho, case):
    q = resolve_q(ho, case["q"])
    a, b, exp = (resolve(case[k], q) for k in ("a", "b", "out"))
    da = dev(hx, a)
    hx.EltwiseAddMod(da, da, b if isinstance(b, int) else dev(hx, b), len(a), q)
    assert host(hx, da).tolist() == exp


@pytest.mark.parametrize("case", KAT["eltwise_sub_mod"]["cases"])
def test_sub_mod_kat(hx, ho, case):
    q = resolve_q(ho, case["q"])
    a, b, exp = (resolve(case[k], q) for k in ("a", "b", "out"))
    da = dev(hx, a)
    hx.EltwiseSubMod(da, da, b if isinstance(b, int) else dev(hx, b), len(a), q)
    assert host(hx, da).tolist() == exp


@pytest.mark.parametrize("bits", [1, 2, 10, 30, 31, 32, 33, 49, 50, 51, 54, 58, 59, 60, 61])
@pytest.mark.parametrize("n,offset", [(1031, 0), (1031, 1), (1, 0), (4096, 0)])
def test_eltwise_vs_oracle(hx, ho, bits, n, offset):
    """n = 1031 (odd on purpose, test/test-eltwise-fma-mod-avx512.cpp:143-211)
    and an 8-byte-misaligned view exercise the scalar head/tail paths."""
    import torch
    rng = np.random.default_rng(bits * 7 + n)
    q = (int(rng.integers(1 << bits, 1 << (bits + 1), dtype=np.uint64)) | 1) if bits > 1 else 3

    def dv(a):
        t = torch.zeros(n + offset + 3, dtype=torch.int64, device="cuda")
        v = t[offset:offset + n]
        v.copy_(hx.from_numpy(a))
        return v

    def out():
        return torch.zeros(n + offset + 3, dtype=torch.int64, device="cuda")[offset:offset + n]

    for in_mf in (1, 2, 4):
        if in_mf * q >= 1 << 63:
            continue
        a = rng.integers(0, in_mf * q, size=n, dtype=np.uint64)
        b = rng.integers(0, in_mf * q, size=n, dtype=np.uint64)
        r = out()
        hx.EltwiseMultMod(r, dv(a), dv(b), n, q, in_mf)
        assert (host(hx, r) == ho.eltwise_mult_mod(a, b, q, in_mf)).all()
    if bits <= 60:
        for in_mf in (1, 2, 4, 8):
            a = rng.integers(0, in_mf * q, size=n, dtype=np.uint64)
            c = rng.integers(0, in_mf * q, size=n, dtype=np.uint64)
            s = int(rng.integers(0, in_mf * q, dtype=np.uint64))
            r = out()
            hx.EltwiseFMAMod(r, dv(a), s, dv(c), n, q, in_mf)
            assert (host(hx, r) == ho.eltwise_fma_mod(a, s, c, q, in_mf)).all()
            hx.EltwiseFMAMod(r, dv(a), s, None, n, q, in_mf)
            assert (host(hx, r) == ho.eltwise_fma_mod(a, s, None, q, in_mf)).all()
        # fused reduce + fma on arbitrary 64-bit words
        a = rng.integers(0, 1 << 64, size=n, dtype=np.uint64)
        c = rng.integers(0, 1 << 64, size=n, dtype=np.uint64)
        s = int(rng.integers(0, q, dtype=np.uint64))
        r = out()
        hx.EltwiseReduceFMAMod(r, dv(a), s, dv(c), n, q, q)
        ref = ho.eltwise_fma_mod(ho.eltwise_reduce_mod(a, q, q, 1), s,
                                 ho.eltwise_reduce_mod(c, q, q, 1), q, 1)
        assert (host(hx, r) == ref).all()
    a = rng.integers(0, q, size=n, dtype=np.uint64)
    b = rng.integers(0, q, size=n, dtype=np.uint64)
    s = int(b[0])
    r = out()
    hx.EltwiseAddMod(r, dv(a), dv(b), n, q)
    assert (host(hx, r) == ho.eltwise_add_mod(a, b, q)).all()
    hx.EltwiseAddMod(r, dv(a), s, n, q)
    assert (host(hx, r) == ho.eltwise_add_mod(a, s, q)).all()
    hx.EltwiseSubMod(r, dv(a), dv(b), n, q)
    assert (host(hx, r) == ho.eltwise_sub_mod(a, b, q)).all()
    hx.EltwiseSubMod(r, dv(a), s, n, q)
    assert (host(hx, r) == ho.eltwise_sub_mod(a, s, q)).all()
    big = rng.integers(0, 1 << 64, size=n, dtype=np.uint64)
    hx.EltwiseReduceMod(r, dv(big), n, q, q, 1)
    assert (host(hx, r) == ho.eltwise_reduce_mod(big, q, q, 1)).all()
    hx.EltwiseReduceMod(r, dv(big), n, q, q, 2)
    got = host(hx, r)
    assert (got < 2 * q).all() and (got % np.uint64(q) == big % np.uint64(q)).all()
    if 4 * q < 1 << 64:
        x4 = rng.integers(0, 4 * q, size=n, dtype=np.uint64)
        hx.EltwiseReduceMod(r, dv(x4), n, q, 4, 1)
        assert (host(hx, r) == ho.eltwise_reduce_mod(x4, q, 4, 1)).all()
        hx.EltwiseReduceMod(r, dv(x4), n, q, 4, 2)
        assert (host(hx, r) == ho.eltwise_reduce_mod(x4, q, 4, 2)).all()
    x2 = rng.integers(0, 2 * q, size=n, dtype=np.uint64)
    hx.EltwiseReduceMod(r, dv(x2), n, q, 2, 1)
    assert (host(hx, r) == ho.eltwise_reduce_mod(x2, q, 2, 1)).all()


def test_eltwise_config2_and_config5_shapes(hx, ho):
    """BASELINE configs[1] (N=4096 x 256, 50-bit) MultMod and configs[4]
    (N=131072 x 1024, 61-bit... FMAMod needs q < 2^61: the first 61-bit prime
    of the survey) at full size, checked on sampled slices + a checksum."""
    import torch
    q = KAT["generate_primes_survey_probe"]["cases"][0]["out"][0]
    n = 4096 * 256
    a = torch.empty(n, dtype=torch.int64, device="cuda")
    b = torch.empty(n, dtype=torch.int64, device="cuda")
    hx.fill_splitmix(a, 4096, 256, 11, q)
    hx.fill_splitmix(b, 4096, 256, 911, q)
    r = torch.empty_like(a)
    hx.EltwiseMultMod(r, a, b, n, q, 1)
    assert (host(hx, r) == ho.eltwise_mult_mod(host(hx, a), host(hx, b), q, 1)).all()

    q = KAT["generate_primes_survey_probe"]["cases"][2]["out"][0]
    N, B = 131072, 1024
    n = N * B
    a = torch.empty(n, dtype=torch.int64, device="cuda")
    c = torch.empty(n, dtype=torch.int64, device="cuda")
    hx.fill_splitmix(a, N, B, 21, 4 * q)
    hx.fill_splitmix(c, N, B, 1021, 4 * q)
    s = 3 * q + 12345
    r = torch.empty_like(a)
    hx.EltwiseFMAMod(r, a, s, c, n, q, 4)
    assert int(r.min()) >= 0 and int(r.max()) < q
    for blk in (0, 511, 1023):
        sl = slice(blk * N, (blk + 1) * N)
        ref = ho.eltwise_fma_mod(host(hx, a[sl]), s, host(hx, c[sl]), q, 4)
        assert (host(hx, r[sl]) == ref).all()
    # fused reduce+fma == reduce, reduce, fma
    hx.fill_splitmix(a, N, B, 31, 0)
    hx.fill_splitmix(c, N, B, 1031, 0)
    fused = torch.empty_like(a)
    hx.EltwiseReduceFMAMod(fused, a, s % q, c, n, q, q)
    ra, rc = torch.empty_like(a), torch.empty_like(a)
    hx.EltwiseReduceMod(ra, a, n, q, q, 1)
    hx.EltwiseReduceMod(rc, c, n, q, q, 1)
    hx.EltwiseFMAMod(ra, ra, s % q, rc, n, q, 1)
    assert torch.equal(fused, ra)
    # ... and the fused kernel against the oracle itself (ReduceMod(q -> 1) of both operands,
    # eltwise-reduce-mod.cpp:32-55, then FMAMod, eltwise-fma-mod-internal.hpp:12-39) on the same
    # three 131072-word blocks as the unfused check above
    for blk in (0, 511, 1023):
        sl = slice(blk * N, (blk + 1) * N)
        oa = ho.eltwise_reduce_mod(host(hx, a[sl]), q, q, 1)
        oc = ho.eltwise_reduce_mod(host(hx, c[sl]), q, q, 1)
        assert (host(hx, fused[sl]) == ho.eltwise_fma_mod(oa, s % q, oc, q, 1)).all()


def test_eltwise_argument_errors(hx):
    import torch
    x = torch.zeros(8, dtype=torch.int64, device="cuda")
    with pytest.raises(hx.HexlAmdError):
        hx.EltwiseMultMod(x, x, x, 8, 769, 3)
    with pytest.raises(hx.HexlAmdError):
        hx.EltwiseFMAMod(x, x, 1, None, 8, 1 << 61, 1)
    with pytest.raises(hx.HexlAmdError):
        hx.EltwiseAddMod(x, x, 800, 8, 769)
    with pytest.raises(hx.HexlAmdError):
        hx.EltwiseReduceMod(x, x, 8, 769, 3, 1)
    with pytest.raises(hx.HexlAmdError):
        hx.EltwiseAddMod(x, x, x, 0, 769)


# ---------------------------------------------------------------- EltwiseCmpAdd / EltwiseCmpSubMod
@pytest.mark.parametrize("case", KAT["eltwise_cmp_add"]["cases"], ids=lambda c: c["cmp"])
def test_cmp_add_kat(hx, ho, case):
    """TEST_P(EltwiseCmpAddTest, Native), test/test-eltwise-cmp-add.cpp:45-78 (in place)."""
    da = dev(hx, case["a"])
    hx.EltwiseCmpAdd(da, da, len(case["a"]), ho.CMPINT[case["cmp"]], case["bound"], case["diff"])
    assert host(hx, da).tolist() == case["out"]


@pytest.mark.parametrize("case", KAT["eltwise_cmp_sub_mod"]["cases"],
                         ids=lambda c: f"{c['cmp']}_{c['q']}")
def test_cmp_sub_mod_kat(hx, ho, case):
    """TEST_P(EltwiseCmpSubModTest, Native), test/test-eltwise-cmp-sub-mod.cpp:49-88."""
    da = dev(hx, case["a"])
    hx.EltwiseCmpSubMod(da, da, len(case["a"]), case["q"], ho.CMPINT[case["cmp"]], case["bound"],
                        case["diff"])
    assert host(hx, da).tolist() == case["out"]


@pytest.mark.parametrize("cmp", range(8))
def test_cmp_ops_random_vs_oracle(hx, ho, cmp):
    """The reference's randomized cross-checks (test/test-eltwise-cmp-add-avx512.cpp:22-52,
    test-eltwise-cmp-sub-mod-avx512.cpp:54-89): lengths 1025 / 172, moduli 100 and 48..51-bit
    primes; plus arbitrary 64-bit words, composite and > 2^62 moduli, misaligned and
    out-of-place buffers."""
    import torch
    rng = np.random.default_rng(100 + cmp)
    for trial in range(6):
        m = 100
        a = rng.integers(0, m, 1025, dtype=np.uint64)
        bound, diff = int(rng.integers(0, m)), int(rng.integers(1, m))
        da = dev(hx, a)
        out = torch.zeros_like(da)
        hx.EltwiseCmpAdd(out, da, a.size, cmp, bound, diff)
        assert np.array_equal(host(hx, out), ho.eltwise_cmp_add(a, cmp, bound, diff))
    for bits in (48, 49, 50, 51):
        m = ho.generate_primes(1, bits, True, 1024)[0]
        a = rng.integers(0, m, 172, dtype=np.uint64)
        bound, diff = int(rng.integers(0, m)), int(rng.integers(1, m - 1))
        da = dev(hx, a)
        hx.EltwiseCmpSubMod(da, da, a.size, m, cmp, bound, diff)
        assert np.array_equal(host(hx, da), ho.eltwise_cmp_sub_mod(a, m, cmp, bound, diff))
    for m in (2, 10, 4294967296, 1152921504606748673, (1 << 62) + 135, (1 << 63) + 29):
        a = rng.integers(0, 1 << 64, 1031, dtype=np.uint64)
        a[:4] = [0, m - 1, m, (1 << 64) - 1]
        bound = int(a[9])
        diff = 1 if m == 2 else int(rng.integers(1, m - 1, dtype=np.uint64))
        buf = dev(hx, np.concatenate([U([0]), a]))  # operand starts 8 bytes off 16-byte alignment
        src = buf[1:]
        out = torch.zeros(a.size + 1, dtype=src.dtype, device=src.device)[1:]
        hx.EltwiseCmpSubMod(out, src, a.size, m, cmp, bound, diff)
        assert np.array_equal(host(hx, out), ho.eltwise_cmp_sub_mod(a, m, cmp, bound, diff))
        hx.EltwiseCmpAdd(out, src, a.size, cmp, bound, diff)
        assert np.array_equal(host(hx, out), ho.eltwise_cmp_add(a, cmp, bound, diff))


def test_cmp_ops_reject_bad_arguments(hx):
    """HEXL_CHECKs of eltwise-cmp-add.cpp:18-21 / eltwise-cmp-sub-mod.cpp:21-25,52-57."""
    da = dev(hx, [1, 2, 3, 4])
    with pytest.raises(hx.HexlAmdError):
        hx.EltwiseCmpAdd(da, da, 0, 0, 1, 1)            # n == 0
    with pytest.raises(hx.HexlAmdError):
        hx.EltwiseCmpAdd(da, da, 4, 0, 1, 0)            # diff == 0
    with pytest.raises(hx.HexlAmdError):
        hx.EltwiseCmpSubMod(da, da, 4, 1, 0, 1, 1)      # modulus <= 1
    with pytest.raises(hx.HexlAmdError):
        hx.EltwiseCmpSubMod(da, da, 4, 10, 0, 1, 10)    # diff >= modulus
    with pytest.raises(hx.HexlAmdError):
        hx.EltwiseCmpSubMod(da, da, 4, 10, 8, 1, 1)     # not a CMPINT


# ---------------------------------------------------------------- DyadicMultiply
@pytest.mark.parametrize("case", KAT["dyadic_multiply"]["cases"], ids=lambda c: c["name"])
def test_dyadic_multiply_kat(hx, case):
    """test/experimental/seal/test-dyadic-multiply.cpp:16-180 incl. the in-place forms."""
    import torch
    op1 = U(case["op1"])
    if case["inplace"]:
        op1 = np.concatenate([op1, np.zeros(op1.size // 2, dtype=np.uint64)])
    d1 = dev(hx, op1)
    d2 = d1 if case["same_op"] else dev(hx, case["op2"])
    out = d1 if case["inplace"] else torch.zeros(len(case["out"]), dtype=d1.dtype, device=d1.device)
    hx.DyadicMultiply(out, d1, d2, case["n"], case["moduli"])
    assert host(hx, out).tolist() == case["out"]


@pytest.mark.parametrize("n,num_moduli", [(4, 1), (512, 3), (4096, 4), (600, 2), (8192, 40)])
def test_dyadic_multiply_random_vs_oracle(hx, ho, n, num_moduli):
    """Sizes around the reference's 512-coefficient tiling (whole tiles only), more moduli
    than one launch carries (40 > 32), RNS primes up to 61 bits."""
    import torch
    rng = np.random.default_rng(n + num_moduli)
    pool = [int(q) for bits in (30, 45, 54, 61) for q in ho.generate_primes(10, bits, True, 8192)]
    moduli = pool[:num_moduli]
    x = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli] * 2)
    y = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli] * 2)
    x[0] = moduli[0] - 1
    y[0] = moduli[0] - 1
    want = ho.dyadic_multiply(x, y, n, moduli, result=np.full(3 * x.size // 2, 5, dtype=np.uint64))
    out = dev(hx, np.full(3 * x.size // 2, 5, dtype=np.uint64))
    hx.DyadicMultiply(out, dev(hx, x), dev(hx, y), n, moduli)
    assert np.array_equal(host(hx, out), want)
    # in place over operand1 (extended by the third polynomial)
    ext = np.concatenate([x, np.full(x.size // 2, 5, dtype=np.uint64)])
    buf = dev(hx, ext)
    hx.DyadicMultiply(buf, buf, dev(hx, y), n, moduli)
    want_inplace = ext.copy()
    ho.dyadic_multiply(want_inplace, y, n, moduli, result=want_inplace)
    assert np.array_equal(host(hx, buf), want_inplace)


@pytest.mark.parametrize("n,num_moduli", [(4, 1), (512, 3), (600, 2), (1024, 1), (4096, 2), (8192, 1), (8192, 2)])
def test_dyadic_multiply_host_small_calls(hx, ho, n, num_moduli):
    """hexl_amd_dyadic_multiply_host on ordinary host memory, both sides of the bounce limit (round 6: calls of
    up to 512 KiB in + out run on the thread's pinned mapped bounce buffer and end on the polled flag; larger ones are
    staged), coefficients beyond the last whole 512-tile left as the caller's result buffer had them
    (dyadic-multiply-internal.cpp:33-34), with the bounce route on and off."""
    import ctypes as C
    rng = np.random.default_rng(7 * n + num_moduli)
    pool = [int(q) for bits in (30, 45, 54, 61) for q in ho.generate_primes(3, bits, True, 8192)]
    moduli = pool[:num_moduli]
    x = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli] * 2)
    y = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli] * 2)
    want = ho.dyadic_multiply(x, y, n, moduli, result=np.full(3 * x.size // 2, 5, dtype=np.uint64))
    mod = (C.c_uint64 * num_moduli)(*moduli)
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    try:
        for kb in (512, 0):
            hx.set_tuning("host_bounce_kb", kb)
            out = np.full(3 * x.size // 2, 5, dtype=np.uint64)
            x0, y0 = x.copy(), y.copy()
            assert hx.lib.hexl_amd_dyadic_multiply_host(p(out), p(x0), p(y0), n, mod, num_moduli) == 0
            assert np.array_equal(out, want), kb
            assert np.array_equal(x0, x) and np.array_equal(y0, y)
    finally:
        hx.set_tuning("host_bounce_kb", 512)


def test_dyadic_multiply_rejects_bad_arguments(hx):
    d = dev(hx, [1, 2, 3, 4, 5, 6, 0, 0, 0])
    with pytest.raises(hx.HexlAmdError):
        hx.DyadicMultiply(d, d, d, 0, [10])
    with pytest.raises(hx.HexlAmdError):
        hx.DyadicMultiply(d, d, d, 3, [1])
    with pytest.raises(hx.HexlAmdError):
        hx.DyadicMultiply(d, d, d, 3, [1 << 62])


def test_dyadic_multiply_batch(hx, ho):
    """Several ciphertext pairs per launch against the oracle pair by pair (n = 1000: the
    reference processes whole 512-coefficient tiles only and leaves the tail untouched)."""
    rng = np.random.default_rng(5)
    for n, k, pairs in ((4096, 3, 5), (1000, 2, 3)):
        moduli = [int(q) for q in ho.generate_primes(k, 50, True, 4096)]
        xs = [np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli] * 2)
              for _ in range(pairs)]
        ys = [np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli] * 2)
              for _ in range(pairs)]
        out = dev(hx, np.full(3 * n * k * pairs, 7, dtype=np.uint64))
        hx.DyadicMultiplyBatch(out, dev(hx, np.concatenate(xs)), dev(hx, np.concatenate(ys)),
                               pairs, n, moduli)
        got = host(hx, out).reshape(pairs, -1)
        for p in range(pairs):
            want = ho.dyadic_multiply(xs[p], ys[p], n, moduli,
                                      result=np.full(3 * n * k, 7, dtype=np.uint64))
            assert np.array_equal(got[p], want)


# ---------------------------------------------------------------- KeySwitch
def _run_key_switch(hx, result, target, n, D, K, R, C, moduli, keys, msf):
    d_res = dev(hx, result)
    hx.KeySwitch(d_res, dev(hx, target), n, D, K, R, C, moduli, [dev(hx, k) for k in keys], msf)
    return host(hx, d_res)


@pytest.mark.parametrize("case", KAT["key_switch"]["cases"])
def test_key_switch_kat(hx, case):
    """TEST(KeySwitch, small), test/experimental/seal/test-key-switch.cpp:16-190."""
    n, D, C = case["n"], case["decomp_modulus_size"], case["key_component_count"]
    inp = U(case["input"])
    got = _run_key_switch(hx, inp[:C * D * n], case["t_target"], n, D, case["key_modulus_size"],
                          case["rns_modulus_size"], C, case["moduli"], case["keys"],
                          case["modswitch_factors"])
    assert got.tolist() == case["out"][:C * D * n]
    assert case["input"][C * D * n:] == case["out"][C * D * n:]  # the tail is never written


@pytest.mark.parametrize("n,D,K,C,bits", [(1024, 1, 2, 2, 40), (4096, 3, 4, 2, 50), (16384, 7, 8, 2, 54),
                                          (8192, 4, 6, 3, 60), (2048, 2, 3, 2, 30)])
def test_key_switch_random_vs_oracle(hx, ho, n, D, K, C, bits):
    """Random RNS bases (mixed sizes so that both the reduce and the copy branches of
    key-switch-internal.cpp:77-86 / :159-167 are taken), more key moduli than used."""
    rng = np.random.default_rng(n + D)
    primes = [int(q) for q in ho.generate_primes(K, bits, True, n)]
    if D >= 2:  # one smaller and one larger decomposition modulus than the special prime
        primes[0] = int(ho.generate_primes(1, bits - 5, True, n)[0])
        primes[1] = int(ho.generate_primes(1, min(bits + 1, 60), False, n)[0])
    moduli = primes
    R = D + 1
    target = np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64) for j in range(D)])
    keys = []
    for j in range(D):
        keys.append(np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                                    for _ in range(C) for i in range(K)]))
    msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
    result = np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                             for _ in range(C) for i in range(D)])
    want = ho.key_switch(result, target, n, D, K, R, C, moduli, keys, msf)
    got = _run_key_switch(hx, result, target, n, D, K, R, C, moduli, keys, msf)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,D,K,C,bits,key_bits", [(4096, 3, 4, 2, 40, 64), (2048, 4, 5, 2, 36, 64),
                                                   (4096, 2, 3, 3, 45, 62), (1024, 16, 17, 2, 55, 0),
                                                   (1024, 17, 18, 2, 55, 0), (1024, 3, 4, 2, 58, 0)])
def test_key_switch_accumulator_on_both_sides_of_the_one_step_limit(hx, ho, n, D, K, C, bits, key_bits):
    """The multiply-accumulate reduces a 128-bit sum below 2^(bits(q) + 61) in one generalised Barrett step
    and every other sum like BarrettReduce128 (util/gcc.hpp:20-28; key-switch-internal.cpp:122-130), decided
    per coefficient.  Key words of up to 64 bits (the reference never reduces or checks them: the sum is still
    exact as long as it fits 128 bits) put coefficients on both sides of the limit in one call; key_bits = 0:
    in-range keys with the largest decomposition counts and moduli around the limit."""
    rng = np.random.default_rng(n + D + bits)
    moduli = [int(q) for q in ho.generate_primes(K, bits, True, n)]
    R = D + 1
    target = np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64) for j in range(D)])
    keys = []
    for j in range(D):
        if key_bits:  # a mixture of magnitudes: log-uniform widths up to key_bits
            width = rng.integers(bits - 4, key_bits + 1, C * K * n)
            k = rng.integers(0, 2**63, C * K * n, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
            keys.append(k >> (np.uint64(64) - width.astype(np.uint64)))
        else:
            keys.append(np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                                        for _ in range(C) for i in range(K)]))
            keys[-1][::3] = np.tile(np.concatenate([np.full(n, moduli[i] - 1, dtype=np.uint64)
                                                    for i in range(K)]), C)[::3]
    msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
    result = np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                             for _ in range(C) for i in range(D)])
    want = ho.key_switch(result, target, n, D, K, R, C, moduli, keys, msf)
    try:
        for onestep in (1, 0):
            hx.set_tuning("ks_mac_onestep", onestep)
            got = _run_key_switch(hx, result, target, n, D, K, R, C, moduli, keys, msf)
            assert np.array_equal(got, want), onestep
    finally:
        hx.set_tuning("ks_mac_onestep", 1)


@pytest.mark.parametrize("n,D,K,C,T,bits", [(4096, 3, 4, 2, 5, 50), (8192, 4, 5, 2, 3, 54),
                                            (64, 2, 3, 2, 4, 40), (16384, 7, 8, 2, 2, 45)])
def test_key_switch_batch_vs_oracle(hx, ho, n, D, K, C, T, bits):
    """Many ciphertexts with the same keys in one call (one sequence of at most eleven launches, the
    per-modulus transforms of all targets in multi-plan NTT launches) against the oracle
    run target by target; moduli of mixed size, hence of mixed arithmetic policy."""
    rng = np.random.default_rng(n + D + T)
    moduli = [int(q) for q in ho.generate_primes(K, bits, True, n)]
    if D >= 2:
        moduli[0] = int(ho.generate_primes(1, bits - 8, True, n)[0])
        moduli[1] = int(ho.generate_primes(1, min(bits + 6, 60), False, n)[0])
    R = D + 1
    keys = [np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                            for _ in range(C) for i in range(K)]) for j in range(D)]
    msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
    targets = [np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64) for j in range(D)])
               for _ in range(T)]
    results = [np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                               for _ in range(C) for i in range(D)]) for _ in range(T)]
    want = np.concatenate([ho.key_switch(results[t], targets[t], n, D, K, R, C, moduli, keys, msf)
                           for t in range(T)])
    d_res = dev(hx, np.concatenate(results))
    hx.KeySwitchBatch(d_res, dev(hx, np.concatenate(targets)), T, n, D, K, R, C, moduli,
                      [dev(hx, k) for k in keys], msf)
    assert np.array_equal(host(hx, d_res), want)


@pytest.mark.parametrize("n,D,K,C,T,bits", [(32768, 3, 4, 2, 3, 54), (8192, 2, 3, 3, 40, 54), (16384, 3, 4, 2, 60, 59),
                                            (4096, 2, 3, 2, 7, 44)])
def test_key_switch_fused_tail(hx, ho, n, D, K, C, T, bits):
    """Round 6: the rounding stage rides on the load and the finish stage on the store of the forward
    transform between them (`ks_fuse`, key-switch-internal.cpp:146-196) -- two-pass plans (n = 32768: the
    strided pass rounds on load, the tile pass folds into the result), one-kernel plans with enough
    polynomials (n = 8192 / 16384: 240 / 360 of them), three key components, moduli of mixed size
    (reduce and copy branches of the rounding, mixed arithmetic policies in one launch sequence):
    fused == stage by stage bit for bit, both == the oracle on the first, a middle and the last target."""
    rng = np.random.default_rng(n + D + T)
    moduli = [int(q) for q in ho.generate_primes(K, bits, True, n)]
    moduli[0] = int(ho.generate_primes(1, bits - 9, True, n)[0])
    if D >= 2:
        moduli[1] = int(ho.generate_primes(1, min(bits + 5, 60), False, n)[0])
    R = D + 1
    keys = [np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                            for _ in range(C) for i in range(K)]) for j in range(D)]
    msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
    targets = [np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64) for j in range(D)])
               for _ in range(T)]
    results = [np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                               for _ in range(C) for i in range(D)]) for _ in range(T)]
    d_keys = [dev(hx, k) for k in keys]
    got = {}
    try:
        for fuse in (0, 1):
            hx.set_tuning("ks_fuse", fuse)
            d_res = dev(hx, np.concatenate(results))
            hx.KeySwitchBatch(d_res, dev(hx, np.concatenate(targets)), T, n, D, K, R, C, moduli, d_keys, msf)
            got[fuse] = host(hx, d_res).reshape(T, -1)
    finally:
        hx.set_tuning("ks_fuse", 1)
    assert np.array_equal(got[0], got[1])
    for t in sorted({0, T // 2, T - 1}):
        assert np.array_equal(got[1][t], ho.key_switch(results[t], targets[t], n, D, K, R, C, moduli, keys, msf))


def _ks_case(ho, rng, n, D, K, C, bits):
    moduli = [int(q) for q in ho.generate_primes(K, bits, True, n)]
    keys = [np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                            for _ in range(C) for i in range(K)]) for _ in range(D)]
    msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
    return moduli, keys, msf


def _ks_data(rng, moduli, n, D, C):
    target = np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64) for j in range(D)])
    result = np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                             for _ in range(C) for i in range(D)])
    return target, result


@pytest.mark.parametrize("n,D,K,C,bits", [(4096, 3, 4, 2, 50), (16384, 7, 8, 2, 54), (1024, 1, 2, 2, 40)])
def test_key_switch_replayed_from_a_graph(hx, ho, n, D, K, C, bits):
    """One ciphertext per call on a stream of the caller's, the same buffers coming back with new
    DATA each time: eager at first sight, captured at the second, replayed from the third
    (hexl_amd_get_counter says so) -- every call bit-exact against the oracle; two buffer sets
    interleaved; with "ks_graph" 0 the same calls run launch by launch."""
    import torch
    rng = np.random.default_rng(n + 31 * D)
    moduli, keys, msf = _ks_case(ho, rng, n, D, K, C, bits)
    d_keys = [dev(hx, k) for k in keys]
    stream = torch.cuda.Stream()
    sets = []
    for _ in range(2):
        t, r = _ks_data(rng, moduli, n, D, C)
        sets.append((dev(hx, t), dev(hx, r)))
    torch.cuda.synchronize()
    c0 = {k: hx.get_counter(k) for k in ("ks_graph_captures", "ks_graph_replays", "ks_eager")}
    for rep in range(5):
        for d_t, d_r in sets:
            t, r = _ks_data(rng, moduli, n, D, C)
            want = ho.key_switch(r, t, n, D, K, D + 1, C, moduli, keys, msf)
            d_t.copy_(dev(hx, t))
            d_r.copy_(dev(hx, r))
            torch.cuda.synchronize()
            with torch.cuda.stream(stream):
                hx.KeySwitch(d_r, d_t, n, D, K, D + 1, C, moduli, d_keys, msf)
            stream.synchronize()
            assert np.array_equal(host(hx, d_r), want), rep
    c1 = {k: hx.get_counter(k) for k in c0}
    assert c1["ks_eager"] - c0["ks_eager"] == 2          # first sight of each buffer set
    assert c1["ks_graph_captures"] - c0["ks_graph_captures"] == 2
    assert c1["ks_graph_replays"] - c0["ks_graph_replays"] == 6
    hx.set_tuning("ks_graph", 0)
    try:
        d_t, d_r = sets[0]
        t, r = _ks_data(rng, moduli, n, D, C)
        d_t.copy_(dev(hx, t))
        d_r.copy_(dev(hx, r))
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            hx.KeySwitch(d_r, d_t, n, D, K, D + 1, C, moduli, d_keys, msf)
        stream.synchronize()
        assert np.array_equal(host(hx, d_r), ho.key_switch(r, t, n, D, K, D + 1, C, moduli, keys, msf))
        assert hx.get_counter("ks_graph_replays") == c1["ks_graph_replays"]
    finally:
        hx.set_tuning("ks_graph", 1)
    # the graphs hold the scratch address: releasing the scratch drops them, the next calls start over
    assert hx.lib.hexl_amd_release_stream_workspaces(stream.cuda_stream) == 0
    for rep in range(3):
        d_t, d_r = sets[1]
        t, r = _ks_data(rng, moduli, n, D, C)
        d_t.copy_(dev(hx, t))
        d_r.copy_(dev(hx, r))
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            hx.KeySwitch(d_r, d_t, n, D, K, D + 1, C, moduli, d_keys, msf)
        stream.synchronize()
        assert np.array_equal(host(hx, d_r), ho.key_switch(r, t, n, D, K, D + 1, C, moduli, keys, msf))
    assert hx.get_counter("ks_graph_captures") == c1["ks_graph_captures"] + 1
    assert hx.lib.hexl_amd_release_stream_workspaces(stream.cuda_stream) == 0


def test_key_switch_replay_cache_evicts_and_tells_parameter_sets_apart(hx, ho):
    """More buffer sets than the cache holds (eight per stream), each seen three times in a
    round-robin -- least recently used entries go, nothing is ever replayed for the wrong buffers;
    then the same buffers with another modulus-switch factor (a different sequence: the factor is
    a kernel argument) -- and a caller that is itself capturing the stream gets plain launches."""
    import torch
    n, D, K, C = 2048, 2, 3, 2
    rng = np.random.default_rng(99)
    moduli, keys, msf = _ks_case(ho, rng, n, D, K, C, 45)
    d_keys = [dev(hx, k) for k in keys]
    stream = torch.cuda.Stream()
    sets = []
    for _ in range(10):
        t, r = _ks_data(rng, moduli, n, D, C)
        sets.append((t, r, dev(hx, t), dev(hx, r)))
    torch.cuda.synchronize()
    for rep in range(3):
        for t, r, d_t, d_r in sets:
            d_r.copy_(dev(hx, r))
            torch.cuda.synchronize()
            with torch.cuda.stream(stream):
                hx.KeySwitch(d_r, d_t, n, D, K, D + 1, C, moduli, d_keys, msf)
            stream.synchronize()
            assert np.array_equal(host(hx, d_r), ho.key_switch(r, t, n, D, K, D + 1, C, moduli, keys, msf))
    t, r, d_t, d_r = sets[-1]
    for other in ([msf[0] + 1] + msf[1:], msf, [msf[0] + 1] + msf[1:], [msf[0] + 1] + msf[1:]):
        d_r.copy_(dev(hx, r))
        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            hx.KeySwitch(d_r, d_t, n, D, K, D + 1, C, moduli, d_keys, other)
        stream.synchronize()
        assert np.array_equal(host(hx, d_r), ho.key_switch(r, t, n, D, K, D + 1, C, moduli, keys, other))
    # inside the caller's own capture the calls are plain launches (a graph launch cannot be captured
    # into another capture as a replay of ours); the caller's graph then replays them
    d_r.copy_(dev(hx, r))
    torch.cuda.synchronize()
    before = hx.get_counter("ks_eager")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=stream):
        hx.KeySwitch(d_r, d_t, n, D, K, D + 1, C, moduli, d_keys, msf)
    assert hx.get_counter("ks_eager") == before + 1
    d_r.copy_(dev(hx, r))
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(host(hx, d_r), ho.key_switch(r, t, n, D, K, D + 1, C, moduli, keys, msf))
    del g
    assert hx.lib.hexl_amd_release_stream_workspaces(stream.cuda_stream) == 0


def test_key_switch_host_calls_replay(hx, ho):
    """hexl_amd_key_switch_host stages into the calling thread's own buffers, so an unmodified
    per-ciphertext loop over host memory replays from its third call; every call against the
    oracle, host keys and device-resident keys."""
    import ctypes as C_
    n, D, K, C = 4096, 3, 4, 2
    rng = np.random.default_rng(5)
    moduli, keys, msf = _ks_case(ho, rng, n, D, K, C, 50)
    mod = (C_.c_uint64 * K)(*moduli)
    fac = (C_.c_uint64 * D)(*msf)
    d_keys = [dev(hx, k) for k in keys]
    for key_ptrs in ([k.ctypes.data for k in keys], [k.data_ptr() for k in d_keys]):
        kp = (C_.c_void_p * D)(*key_ptrs)
        r0 = hx.get_counter("ks_graph_replays")
        for rep in range(5):
            t, r = _ks_data(rng, moduli, n, D, C)
            want = ho.key_switch(r, t, n, D, K, D + 1, C, moduli, keys, msf)
            got = r.copy()
            rc = hx.lib.hexl_amd_key_switch_host(got.ctypes.data_as(C_.c_void_p),
                                                 t.ctypes.data_as(C_.c_void_p), n, D, K, D + 1, C, mod,
                                                 C_.cast(kp, C_.POINTER(C_.c_void_p)), fac)
            assert rc == 0, hx.lib.hexl_amd_last_error()
            assert np.array_equal(got, want), rep
        assert hx.get_counter("ks_graph_replays") >= r0 + 3


def test_workspaces_can_be_released(hx, ho):
    """Scratch of the composites is cached per (device, stream); the release entry points give
    it back, and a later call simply allocates again."""
    import torch
    n, D, K, C = 4096, 2, 3, 2
    rng = np.random.default_rng(9)
    moduli = [int(q) for q in ho.generate_primes(K, 50, True, n)]
    keys = [np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                            for _ in range(C) for i in range(K)]) for _ in range(D)]
    msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
    target = np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64) for j in range(D)])
    result = np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                             for _ in range(C) for i in range(D)])
    want = ho.key_switch(result, target, n, D, K, D + 1, C, moduli, keys, msf)
    d_keys = [dev(hx, k) for k in keys]
    st = torch.cuda.Stream()
    for rep in range(3):
        with torch.cuda.stream(st):
            o = dev(hx, result)
            hx.KeySwitch(o, dev(hx, target), n, D, K, D + 1, C, moduli, d_keys, msf)
        st.synchronize()
        assert np.array_equal(host(hx, o), want)
        if rep == 0:
            assert hx.lib.hexl_amd_release_stream_workspaces(st.cuda_stream) == 0
        elif rep == 1:
            torch.cuda.synchronize()
            assert hx.lib.hexl_amd_release_workspaces() == 0


def test_key_switch_host_with_device_resident_keys(hx, ho):
    """hexl_amd_key_switch_host: host result / target; key blocks on the host (copied per call), on
    the device (uploaded once, used where they lie) or mixed -- the same bits as the oracle."""
    import ctypes as C
    import time
    n, D, K, C_ = 8192, 4, 5, 2
    rng = np.random.default_rng(31)
    moduli = [int(q) for q in ho.generate_primes(K, 54, True, n)]
    keys = [np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                            for _ in range(C_) for i in range(K)]) for _ in range(D)]
    msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
    target = np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64) for j in range(D)])
    result = np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                             for _ in range(C_) for i in range(D)])
    want = ho.key_switch(result, target, n, D, K, D + 1, C_, moduli, keys, msf)
    mod = (C.c_uint64 * K)(*moduli)
    fac = (C.c_uint64 * D)(*msf)
    d_keys = [dev(hx, k) for k in keys]
    p = lambda a: a.ctypes.data_as(C.c_void_p).value  # noqa: E731
    times = {}
    for label, ptrs in (("host keys", [p(k) for k in keys]),
                        ("device keys", [k.data_ptr() for k in d_keys]),
                        ("mixed", [p(keys[0]), d_keys[1].data_ptr(), p(keys[2]), d_keys[3].data_ptr()])):
        kp = (C.c_void_p * D)(*ptrs)
        for rep in range(3):
            r = result.copy()
            t0 = time.perf_counter()
            rc = hx.lib.hexl_amd_key_switch_host(r.ctypes.data_as(C.c_void_p),
                                                 target.ctypes.data_as(C.c_void_p), n, D, K, D + 1,
                                                 C_, mod, C.cast(kp, C.POINTER(C.c_void_p)), fac)
            times[label] = time.perf_counter() - t0
            assert rc == 0, hx.lib.hexl_amd_last_error()
            assert np.array_equal(r, want), label
    print("KeySwitch from host buffers, n=8192 D=4:", {k: f"{v * 1e6:.0f} us" for k, v in times.items()})


def test_key_switch_two_streams_do_not_share_scratch(hx, ho):
    """KeySwitch calls issued from one thread on two streams overlap on the device; their
    scratch (coefficient-form targets, operand transforms, products) is keyed by stream, so
    both results are right (round 1 kept one scratch per thread: advisor finding)."""
    import torch
    n, D, K, C = 8192, 3, 4, 2
    rng = np.random.default_rng(17)
    moduli = [int(q) for q in ho.generate_primes(K, 54, True, n)]
    keys = [np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                            for _ in range(C) for i in range(K)]) for _ in range(D)]
    msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
    d_keys = [dev(hx, k) for k in keys]
    cases = []
    for _ in range(2):
        target = np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64) for j in range(D)])
        result = np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                                 for _ in range(C) for i in range(D)])
        want = ho.key_switch(result, target, n, D, K, D + 1, C, moduli, keys, msf)
        cases.append((dev(hx, target), dev(hx, result), want))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    outs = [[], []]
    for rep in range(6):  # interleaved issue: the two streams' launches alternate
        for i, (d_t, d_r, want) in enumerate(cases):
            with torch.cuda.stream(streams[i]):
                o = d_r.clone()
                hx.KeySwitch(o, d_t, n, D, K, D + 1, C, moduli, d_keys, msf)
                outs[i].append(o)
    torch.cuda.synchronize()
    for i, (d_t, d_r, want) in enumerate(cases):
        for o in outs[i]:
            assert np.array_equal(host(hx, o), want)


def test_key_switch_two_threads_one_stream(hx, ho):
    """Two host threads issuing KeySwitch on the SAME stream (both on the default stream):
    each call's launches must stay together on the stream, or the two calls run over
    each other's scratch (the sequence lock of workspace.h)."""
    import threading
    import torch
    n, D, K, C = 4096, 3, 4, 2
    rng = np.random.default_rng(23)
    moduli = [int(q) for q in ho.generate_primes(K, 52, True, n)]
    keys = [np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                            for _ in range(C) for i in range(K)]) for _ in range(D)]
    msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
    d_keys = [dev(hx, k) for k in keys]
    cases = []
    for _ in range(2):
        target = np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64) for j in range(D)])
        result = np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                                 for _ in range(C) for i in range(D)])
        want = ho.key_switch(result, target, n, D, K, D + 1, C, moduli, keys, msf)
        cases.append((dev(hx, target), dev(hx, result), want))
    torch.cuda.synchronize()
    outs, errors = [[], []], []
    start = threading.Barrier(2)

    def work(i):
        try:
            d_t, d_r, _ = cases[i]
            start.wait()
            for _ in range(40):
                o = d_r.clone()
                hx.KeySwitch(o, d_t, n, D, K, D + 1, C, moduli, d_keys, msf)
                outs[i].append(o)
        except Exception as exc:  # noqa: BLE001 - reported below
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for i in range(2):
        for o in outs[i]:
            assert np.array_equal(host(hx, o), cases[i][2])


def test_key_switch_rejects_bad_arguments(hx, ho):
    n = 16
    q = [int(x) for x in ho.generate_primes(3, 40, True, n)]
    d = dev(hx, np.zeros(4 * n, dtype=np.uint64))
    with pytest.raises(hx.HexlAmdError):   # rns_modulus_size != decomp + 1
        hx.KeySwitch(d, d, n, 2, 3, 2, 2, q, [d, d], [1, 1])
    with pytest.raises(hx.HexlAmdError):   # composite modulus
        hx.KeySwitch(d, d, n, 2, 3, 3, 2, [q[0], 1000, q[2]], [d, d], [1, 1])


# ---------------------------------------------------------------- maximum degree
@pytest.mark.parametrize("log_n,bits", [(18, 54), (19, 61), (20, 54), (20, 29), (19, 49), (18, 59)])
def test_ntt_maximum_degrees(hx, ho, log_n, bits):
    """N up to 2^20 = NTT::MaxDegreeBits() (hexl/include/hexl/ntt/ntt.hpp:197), all five
    arithmetic policies: 2^18 / 2^19 as five strided stages + the 13- / 14-stage tile pass of the
    one-kernel plans, 2^20 as two strided passes in front of a 12-stage tile pass."""
    n = 1 << log_n
    q = ho.generate_primes(1, bits, True, n)[0]
    x = ho.fill_splitmix(n, log_n * 31 + bits, q)
    want = ho.NTT(n, q).forward(x, 1, 1)
    gnt = hx.NTT(n, q)
    d = dev(hx, x)
    gnt.ComputeForward(d, d, 1, 1)
    assert np.array_equal(host(hx, d), want)
    gnt.ComputeInverse(d, d, 1, 1)
    assert np.array_equal(host(hx, d), x)


@pytest.mark.parametrize("bigtile", [1, 0])
@pytest.mark.parametrize("log_n,bits", [(18, 54), (19, 54), (19, 28), (18, 49), (19, 60)])
def test_ntt_big_degrees_both_plans(hx, ho, log_n, bits, bigtile):
    """N = 2^18, 2^19 under both plans the library holds for them -- set_tuning("bigtile", 1):
    5 strided stages + one 13- / 14-stage tile pass (two HBM round trips); 0: three passes
    (3 + 3 + 12, 4 + 3 + 12) -- several polynomials per call, every legal (input, output) factor
    pair, against the oracle (round-3 advisor: the big-tile plan had only (1, 1) on <= 4
    polynomials, the three-pass plan no test at all once the default changed)."""
    n = 1 << log_n
    q = ho.generate_primes(1, bits, True, n)[0]
    ont, batch = ho.NTT(n, q), 5
    try:
        hx.set_tuning("bigtile", bigtile)
        gnt = hx.NTT(n, q)
        for in_mf, out_mf in ((1, 1), (4, 4), (2, 1)):
            x = np.stack([ho.fill_splitmix(n, 900 + 10 * log_n + b, in_mf * q) for b in range(batch)])
            d = dev(hx, x)
            gnt.ComputeForward(d, d, in_mf, out_mf)
            got = host(hx, d)
            for b in (0, batch - 1):  # (the oracle takes ~0.1 s per polynomial here)
                want = ont.forward(x[b], in_mf, 1)
                assert (got[b] < np.uint64(out_mf * q)).all() and np.array_equal(got[b] % np.uint64(q), want)
        for in_mf, out_mf in ((1, 1), (2, 2), (2, 1)):
            x = np.stack([ho.fill_splitmix(n, 950 + 10 * log_n + b, in_mf * q) for b in range(batch)])
            d = dev(hx, x)
            out = dev(hx, np.zeros_like(x))
            gnt.ComputeInverse(out, d, in_mf, out_mf)  # out of place
            got = host(hx, out)
            for b in (0, batch - 1):
                want = ont.inverse(x[b], in_mf, 1)
                assert (got[b] < np.uint64(out_mf * q)).all() and np.array_equal(got[b] % np.uint64(q), want)
    finally:
        hx.set_tuning("bigtile", 1)
