"""Pins the CPU oracle (oracle/hexl_oracle.c) against every known-answer vector
the reference's own tests hold for the NTT / Eltwise path
(tests/golden/hexl_kat.json; each block cites the reference test file:line),
against an independent big-integer evaluation of the transform's definition,
and against the cross-implementation properties of test/test-ntt.cpp:406-478.
CPU only.
"""
import json
import os

import numpy as np
import pytest

from oracle import hexl_oracle as ho

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden",
                                  "hexl_kat.json")))


def resolve_q(q):
    if isinstance(q, dict):
        return ho.generate_primes(*q["gp"])[0]
    return q


def resolve(v, q):
    """Fixture values may be ints or {"q_minus": k} (= modulus - k)."""
    if isinstance(v, dict):
        return q - v["q_minus"]
    if isinstance(v, list):
        return [resolve(x, q) for x in v]
    return v


U = lambda x: np.asarray(x, dtype=np.uint64)


# ---------------------------------------------------------------- NTT KATs
@pytest.mark.parametrize("case", KAT["ntt_forward"]["cases"],
                         ids=lambda c: f"n{c['n']}_q{c['q']}")
def test_ntt_kat_api(case):
    """Mirrors TEST_P(DegreeModulusInputOutput, API), test/test-ntt.cpp:227-339."""
    n, q = case["n"], case["q"]
    inp, exp = U(case["in"]), U(case["out"])
    ntt = ho.NTT(n, q)

    # in-place forward, canonical
    buf = inp.copy()
    ntt.forward_inplace(buf, 1, 1)
    assert (buf == exp).all()
    # in-place lazy forward: compared mod q (test-ntt.cpp:246-251)
    buf = inp.copy()
    ntt.forward_inplace(buf, 2, 4)
    assert (buf < 4 * q).all()
    assert (buf % np.uint64(q) == exp).all()
    # reference (fully reduced) forward and inverse
    ref = ntt.forward_reference(inp)
    assert (ref == exp).all()
    assert (ntt.inverse_reference(ref) == inp).all()
    # out-of-place round trip
    out = ntt.forward(inp, 1, 1)
    assert (out == exp).all()
    assert (ntt.inverse(out, 1, 1) == inp).all()
    # out-of-place forward with in_mf = 2
    assert (ntt.forward(inp, 2, 1) == exp).all()
    # lazy inverse compared mod q (test-ntt.cpp:279-286)
    lazy = ntt.inverse(exp, 1, 2)
    assert (lazy < 2 * q).all()
    assert (lazy % np.uint64(q) == inp).all()
    # in-place inverse
    buf = exp.copy()
    ntt.inverse_inplace(buf, 1, 1)
    assert (buf == inp).all()


@pytest.mark.parametrize("case", KAT["ntt_root_powers"]["cases"])
def test_ntt_root_powers(case):
    ntt = ho.NTT(case["n"], case["q"])
    assert [int(x) for x in ntt.root_pows] == case["powers"]


def test_minimal_root_probe():
    for c in KAT["ntt_minimal_root_survey_probe"]["cases"]:
        assert ho.NTT(c["n"], c["q"]).w == c["w"]


def test_inverse_table_layout():
    """hexl/ntt/ntt-internal.cpp:143-154: IR[0]=1, then inverses of
    R[m..2m-1] for m = N/2, N/4, .., 1; precon = floor(W*2^64/q)."""
    n, q = 64, 769
    ntt = ho.NTT(n, q)
    R = [int(x) for x in ntt.root_pows]
    IR = [int(x) for x in ntt.inv_root_pows]
    assert IR[0] == 1
    idx = 1
    m = n // 2
    while m > 0:
        for i in range(m):
            assert IR[idx] * R[m + i] % q == 1
            idx += 1
        m //= 2
    for k in range(n):
        assert int(ntt.precon_root_pows[k]) == (R[k] << 64) // q
        assert int(ntt.precon_inv_root_pows[k]) == (IR[k] << 64) // q
    # R[bitrev(i)] = w^i
    for i in range(n):
        assert R[int(ho.reverse_bits(i, 6))] == pow(ntt.w, i, q)


@pytest.mark.parametrize("n,q", [(8, 4194353), (64, 769), (256, 0xffffee001),
                                 (1024, 0xffffee001)])
def test_ntt_matches_definition(n, q):
    """Independent oracle: out[i] = sum_j a_j * w^((2*bitrev(i)+1)*j) mod q in
    Python big integers."""
    ntt = ho.NTT(n, q)
    bits = n.bit_length() - 1
    rng = np.random.default_rng(n)
    a = [int(x) for x in rng.integers(0, q, size=n, dtype=np.uint64)]
    got = ntt.forward(U(a), 1, 1)
    w = ntt.w
    rows = range(n) if n <= 256 else rng.integers(0, n, size=48)
    for i in rows:
        e = 2 * int(ho.reverse_bits(int(i), bits)) + 1
        wi = pow(w, e, q)
        acc, p = 0, 1
        for j in range(n):
            acc = (acc + a[j] * p) % q
            p = p * wi % q
        assert int(got[int(i)]) == acc


DEFN = json.load(open(os.path.join(os.path.dirname(__file__), "golden",
                                   "ntt_definition_fixtures.json")))


@pytest.mark.parametrize("case", DEFN["cases"], ids=lambda c: "n%d" % c["n"])
def test_ntt_matches_definition_at_benchmark_sizes(case):
    """The oracle pinned where the benchmark lives (N = 4096 / 65536 / 131072 with the
    survey's primes): 64 forward and 64 inverse output entries computed from the
    transform's definition in Python big integers by
    tests/golden/make_ntt_definition_fixtures.py (independent of the oracle), plus the
    digest of the full vector those entries belong to."""
    import hashlib
    n, q = case["n"], case["q"]
    ntt = ho.NTT(n, q)
    assert ntt.w == case["minimal_root"]
    f = ntt.forward(ho.fill_splitmix(n, case["forward"]["seed"], q), 1, 1)
    for i, v in case["forward"]["samples"]:
        assert int(f[i]) == v
    assert hashlib.sha256(f.astype("<u8").tobytes()).hexdigest() == case["forward"]["sha256_le_u64"]
    b = ntt.inverse(ho.fill_splitmix(n, case["inverse"]["seed"], q), 1, 1)
    for j, v in case["inverse"]["samples"]:
        assert int(b[j]) == v
    assert hashlib.sha256(b.astype("<u8").tobytes()).hexdigest() == case["inverse"]["sha256_le_u64"]


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048,
                               4096, 8192])
@pytest.mark.parametrize("bits", [27, 33, 49, 54, 60])
def test_ntt_radix2_vs_reference(n, bits):
    """Cross-implementation equality on random inputs, the reference's
    pattern-2 tests (test/test-ntt.cpp:406-478, test-ntt-avx512.cpp:169-398)."""
    q = ho.generate_primes(1, bits, True, n)[0]
    ntt = ho.NTT(n, q)
    x = ho.fill_splitmix(n, bits * 131 + n, q)
    f = ntt.forward(x, 1, 1)
    assert (f == ntt.forward_reference(x)).all()
    assert (ntt.inverse(f, 1, 1) == x).all()
    assert (ntt.inverse_reference(f) == x).all()
    # lazy inputs / outputs
    x4 = ho.fill_splitmix(n, n + 7, 4 * q)
    f4 = ntt.forward(x4, 4, 4)
    assert (f4 < 4 * q).all()
    assert (f4 % np.uint64(q) == ntt.forward(x4 % np.uint64(q), 1, 1)).all()
    x2 = ho.fill_splitmix(n, n + 9, 2 * q)
    i2 = ntt.inverse(x2, 2, 2)
    assert (i2 < 2 * q).all()
    assert (i2 % np.uint64(q) == ntt.inverse(x2 % np.uint64(q), 1, 1)).all()


@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64, 128, 256, 1024, 2048, 4096, 8192])
@pytest.mark.parametrize("bits", [27, 49, 60])
def test_ntt_radix4_equals_radix2(n, bits):
    """The reference's second native implementation (hexl/ntt/ntt-radix-4.cpp) restated and
    held against the radix-2 one the way the reference's tests do (test/test-ntt.cpp:318-355,
    :422-452): in place and out of place, every (in, out) factor -- lazy outputs included,
    because a radix-4 butterfly is four radix-2 butterflies and every value coincides."""
    q = ho.generate_primes(1, bits, True, n)[0]
    ntt = ho.NTT(n, q)
    for in_mf, out_mf in ((1, 1), (2, 1), (4, 1), (4, 4), (1, 4)):
        x = ho.fill_splitmix(n, bits + n + in_mf, in_mf * q)
        want = ntt.forward(x, in_mf, out_mf)
        assert (ntt.forward_radix4(x, in_mf, out_mf) == want).all()
        assert (ntt.forward_radix4(x, in_mf, out_mf, inplace=True) == want).all()
    for in_mf, out_mf in ((1, 1), (2, 1), (2, 2), (1, 2)):
        x = ho.fill_splitmix(n, bits + n + 7 * in_mf, in_mf * q)
        want = ntt.inverse(x, in_mf, out_mf)
        assert (ntt.inverse_radix4(x, in_mf, out_mf) == want).all()
        assert (ntt.inverse_radix4(x, in_mf, out_mf, inplace=True) == want).all()


@pytest.mark.parametrize("case", KAT["ntt_forward"]["cases"], ids=lambda c: f"n{c['n']}_q{c['q']}")
def test_ntt_radix4_kat(case):
    """The reference's hand vectors through the radix-4 restatement (test/test-ntt.cpp:318-355)."""
    ntt = ho.NTT(case["n"], case["q"])
    assert ntt.forward_radix4(U(case["in"]), 1, 1).tolist() == case["out"]
    assert ntt.inverse_radix4(U(case["out"]), 1, 1).tolist() == case["in"]


def test_ntt_headline_config_roundtrip():
    """BASELINE.json configs[2]: N=65536, q = first 55-bit prime."""
    n = 65536
    q = KAT["generate_primes_survey_probe"]["cases"][1]["out"][0]
    ntt = ho.NTT(n, q)
    x = ho.fill_splitmix(n, 1, q)
    f = ntt.forward(x, 1, 1)
    assert (f < q).all()
    assert (ntt.inverse(f, 1, 1) == x).all()
    # linearity spot check: NTT(a) + NTT(b) == NTT(a+b)
    y = ho.fill_splitmix(n, 2, q)
    s = ho.eltwise_add_mod(x, y, q)
    assert (ho.eltwise_add_mod(f, ntt.forward(y, 1, 1), q) ==
            ntt.forward(s, 1, 1)).all()


# ---------------------------------------------------------------- eltwise KATs
@pytest.mark.parametrize("case", KAT["eltwise_mult_mod"]["cases"])
def test_mult_mod_kat(case):
    q = resolve_q(case["q"])
    a, b, exp = (resolve(case[k], q) for k in ("a", "b", "out"))
    assert ho.eltwise_mult_mod(a, b, q, case["in_mf"]).tolist() == exp


@pytest.mark.parametrize("case", KAT["eltwise_fma_mod"]["cases"])
def test_fma_mod_kat(case):
    q = resolve_q(case["q"])
    got = ho.eltwise_fma_mod(case["a"], case["s"], case["c"], q, case["in_mf"])
    assert got.tolist() == case["out"]


@pytest.mark.parametrize("case", KAT["eltwise_reduce_mod"]["cases"])
def test_reduce_mod_kat(case):
    q = case["q"]
    in_mf = q if case["in_mf"] == "q" else case["in_mf"]
    got = ho.eltwise_reduce_mod(case["a"], q, in_mf, case["out_mf"])
    assert got.tolist() == case["out"]


@pytest.mark.parametrize("case", KAT["eltwise_add_mod"]["cases"])
def test_add_mod_kat(case):
    q = resolve_q(case["q"])
    a, b, exp = (resolve(case[k], q) for k in ("a", "b", "out"))
    assert ho.eltwise_add_mod(a, b, q).tolist() == exp


@pytest.mark.parametrize("case", KAT["eltwise_sub_mod"]["cases"])
def test_sub_mod_kat(case):
    q = resolve_q(case["q"])
    a, b, exp = (resolve(case[k], q) for k in ("a", "b", "out"))
    assert ho.eltwise_sub_mod(a, b, q).tolist() == exp


@pytest.mark.parametrize("bits", [1, 2, 10, 30, 31, 32, 33, 49, 50, 51, 58, 59,
                                  60, 61])
def test_eltwise_vs_bigint(bits):
    """Random properties with n = 1031 (n % 8 != 0 on purpose, as in
    test/test-eltwise-fma-mod-avx512.cpp:143-211) against Python integers."""
    n = 1031
    rng = np.random.default_rng(bits)
    q = (int(rng.integers(1 << bits, 1 << (bits + 1), dtype=np.uint64)) | 1) if bits > 1 else 3
    for in_mf in (1, 2, 4):
        a = rng.integers(0, in_mf * q, size=n, dtype=np.uint64)
        b = rng.integers(0, in_mf * q, size=n, dtype=np.uint64)
        exp = [int(x) * int(y) % q for x, y in zip(a, b)]
        assert ho.eltwise_mult_mod(a, b, q, in_mf).tolist() == exp
    if bits <= 60:  # EltwiseFMAMod requires q < 2^61 (eltwise-fma-mod.cpp:24)
        for in_mf in (1, 2, 4, 8):
            if in_mf * q >= 1 << 64:
                continue
            a = rng.integers(0, in_mf * q, size=n, dtype=np.uint64)
            c = rng.integers(0, in_mf * q, size=n, dtype=np.uint64)
            s = int(rng.integers(0, in_mf * q, dtype=np.uint64))
            exp = [(int(x) * s + int(z)) % q for x, z in zip(a, c)]
            assert ho.eltwise_fma_mod(a, s, c, q, in_mf).tolist() == exp
            exp = [(int(x) * s) % q for x in a]
            assert ho.eltwise_fma_mod(a, s, None, q, in_mf).tolist() == exp
    a = rng.integers(0, q, size=n, dtype=np.uint64)
    b = rng.integers(0, q, size=n, dtype=np.uint64)
    assert ho.eltwise_add_mod(a, b, q).tolist() == [
        (int(x) + int(y)) % q for x, y in zip(a, b)]
    assert ho.eltwise_sub_mod(a, b, q).tolist() == [
        (int(x) - int(y)) % q for x, y in zip(a, b)]
    s = int(b[0])
    assert ho.eltwise_add_mod(a, s, q).tolist() == [(int(x) + s) % q for x in a]
    assert ho.eltwise_sub_mod(a, s, q).tolist() == [(int(x) - s) % q for x in a]
    big = rng.integers(0, 1 << 64, size=n, dtype=np.uint64)
    assert ho.eltwise_reduce_mod(big, q, q, 1).tolist() == [int(x) % q for x in big]
    r2 = ho.eltwise_reduce_mod(big, q, q, 2)
    assert (r2 < 2 * q).all() and (r2 % np.uint64(q)).tolist() == [int(x) % q for x in big]
    x4 = rng.integers(0, 4 * q, size=n, dtype=np.uint64)
    assert ho.eltwise_reduce_mod(x4, q, 4, 1).tolist() == [int(x) % q for x in x4]
    assert ho.eltwise_reduce_mod(x4, q, 4, 2).tolist() == [
        int(x) - 2 * q if int(x) >= 2 * q else int(x) for x in x4]
    x2 = rng.integers(0, 2 * q, size=n, dtype=np.uint64)
    assert ho.eltwise_reduce_mod(x2, q, 2, 1).tolist() == [int(x) % q for x in x2]


# ---------------------------------------------------------------- number theory
NT = KAT["number_theory"]


def test_nt_multiply_mod():
    for m, x, y, e in NT["multiply_mod"]:
        assert ho.multiply_mod(x, y, m) == e
        pre = ho.multiply_factor(y, 64, m)
        assert ho.multiply_mod_precon(x, y, pre, m) == e


def test_nt_multiply_mod_lazy():
    for m, x, y, e in NT["multiply_mod_lazy64"]:
        pre = ho.multiply_factor(y, 64, m)
        assert ho.multiply_mod_lazy64(x, y, pre, m) == e


def test_nt_misc():
    for m, b, e, r in NT["pow_mod"]:
        assert ho.pow_mod(b, e, m) == r
    for m, root, deg, r in NT["is_primitive_root"]:
        assert ho.is_primitive_root(root, deg, m) == r
    for m, deg, r in NT["minimal_primitive_root"]:
        assert ho.minimal_primitive_root(deg, m) == r
    for x, m, r in NT["inverse_mod"]:
        assert ho.inverse_mod(x, m) == r
    for x, w, r in NT["reverse_bits"]:
        assert ho.reverse_bits(x, w) == r
    for n, r in NT["is_prime"]:
        assert ho.is_prime(n) == r
    for x1, x0, y, r in NT["divide_u128_u64_lo"]:
        assert ho.divide_u128_u64_lo(x1, x0, y) == r
    for x, r in NT["msb"]:
        assert ho.msb(x) == r
    for m, x, y, r in NT["add_uint_mod"]:
        assert ho.add_uint_mod(x, y, m) == r
    for m, x, y, r in NT["sub_uint_mod"]:
        assert ho.sub_uint_mod(x, y, m) == r


def test_nt_generate_primes():
    g = NT["generate_primes_property"]
    for bit_size in range(g["bit_sizes"][0], g["bit_sizes"][1] + 1):
        for small in (True, False):
            ps = ho.generate_primes(g["num"], bit_size, small, g["ntt_size"])
            assert len(ps) == g["num"]
            for p in ps:
                assert p % (2 * g["ntt_size"]) == 1 and ho.is_prime(p)
                assert (1 << bit_size) <= p <= (1 << (bit_size + 1))
    for c in KAT["generate_primes_survey_probe"]["cases"]:
        assert ho.generate_primes(*c["args"]) == c["out"]


# ---------------------------------------------------------------- EltwiseCmpAdd / EltwiseCmpSubMod
@pytest.mark.parametrize("case", KAT["eltwise_cmp_add"]["cases"], ids=lambda c: c["cmp"])
def test_cmp_add_kat(case):
    got = ho.eltwise_cmp_add(case["a"], ho.CMPINT[case["cmp"]], case["bound"], case["diff"])
    assert got.tolist() == case["out"]


@pytest.mark.parametrize("case", KAT["eltwise_cmp_sub_mod"]["cases"],
                         ids=lambda c: f"{c['cmp']}_{c['q']}")
def test_cmp_sub_mod_kat(case):
    got = ho.eltwise_cmp_sub_mod(case["a"], case["q"], ho.CMPINT[case["cmp"]], case["bound"],
                                 case["diff"])
    assert got.tolist() == case["out"]


def test_cmp_ops_against_python_ints():
    """Independent model in Python integers over arbitrary 64-bit words, composite and
    prime moduli (hexl/eltwise/eltwise-cmp-sub-mod.cpp:58-64, eltwise-cmp-add.cpp:40-103)."""
    rng = np.random.default_rng(5)
    py_cmp = [lambda a, b: a == b, lambda a, b: a < b, lambda a, b: a <= b, lambda a, b: False,
              lambda a, b: a != b, lambda a, b: a >= b, lambda a, b: a > b, lambda a, b: True]
    for m in (2, 10, 769, 4294967296, 1152921504606748673, (1 << 62) + 135):
        a = rng.integers(0, 1 << 64, 257, dtype=np.uint64)
        a[:4] = [0, m - 1, m, (1 << 64) - 1]
        bound = int(a[7])
        diff = 1 + int(rng.integers(0, m - 1, dtype=np.uint64)) if m > 2 else 1
        for cmp in range(8):
            got = ho.eltwise_cmp_sub_mod(a, m, cmp, bound, diff).tolist()
            want = [((int(x) % m) - diff) % m if py_cmp[cmp](int(x), bound) else int(x) % m
                    for x in a]
            assert got == want
            got = ho.eltwise_cmp_add(a, cmp, bound, diff).tolist()
            want = [(int(x) + diff) % (1 << 64) if py_cmp[cmp](int(x), bound) else int(x)
                    for x in a]
            assert got == want


# ---------------------------------------------------------------- DyadicMultiply
def _dyadic_buffers(case):
    op1 = np.asarray(case["op1"], dtype=np.uint64)
    op2 = op1 if case["same_op"] else np.asarray(case["op2"], dtype=np.uint64)
    if case["inplace"]:  # result aliases operand1, which is extended by the output polynomial
        op1 = np.concatenate([op1, np.zeros(op1.size // 2, dtype=np.uint64)])
        if case["same_op"]:
            op2 = op1
    return op1, op2


@pytest.mark.parametrize("case", KAT["dyadic_multiply"]["cases"], ids=lambda c: c["name"])
def test_dyadic_multiply_kat(case):
    op1, op2 = _dyadic_buffers(case)
    out = ho.dyadic_multiply(op1, op2, case["n"], case["moduli"],
                             result=op1 if case["inplace"] else None)
    assert out.tolist() == case["out"]


def test_dyadic_multiply_against_python_ints():
    """Definition check with RNS primes and the reference's tiling (n = 2048 is four tiles
    of 512; n = 600 leaves 88 coefficients per modulus untouched,
    dyadic-multiply-internal.cpp:33-34)."""
    rng = np.random.default_rng(9)
    for n in (4, 512, 2048, 600):
        moduli = [int(q) for q in ho.generate_primes(3, 40, True, 1024)] + [10]
        k = len(moduli)
        x = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli] * 2)
        y = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli] * 2)
        res = np.full(3 * n * k, 7, dtype=np.uint64)
        ho.dyadic_multiply(x, y, n, moduli, result=res)
        n_proc = (n // min(n, 512)) * min(n, 512)
        for i, q in enumerate(moduli):
            for e in (0, 1, n_proc - 1, n - 1):
                x0, x1 = int(x[i * n + e]), int(x[n * k + i * n + e])
                y0, y1 = int(y[i * n + e]), int(y[n * k + i * n + e])
                want = ([x0 * y0 % q, (x0 * y1 + x1 * y0) % q, x1 * y1 % q] if e < n_proc
                        else [7, 7, 7])
                assert [int(res[p * n * k + i * n + e]) for p in range(3)] == want


# ---------------------------------------------------------------- KeySwitch
def _key_switch_args(case):
    return (case["n"], case["decomp_modulus_size"], case["key_modulus_size"],
            case["rns_modulus_size"], case["key_component_count"], case["moduli"], case["keys"],
            case["modswitch_factors"])


@pytest.mark.parametrize("case", KAT["key_switch"]["cases"])
def test_key_switch_kat(case):
    """TEST(KeySwitch, small), test/experimental/seal/test-key-switch.cpp:16-190: pins the
    oracle's inverse/forward NTTs with lazy outputs, ReduceMod, the 128-bit accumulation and
    FMAMod(8) in one composite known answer."""
    got = ho.key_switch(case["input"], case["t_target"], *_key_switch_args(case))
    assert got.tolist() == case["out"]


# ---------------------------------------------------------------- AVX-512 baseline variant
def test_avx512_variant_matches_scalar():
    """oracle/hexl_oracle_avx512.c (the cpu_baseline of bench.py) against the scalar
    restatement: every (in_mf, out_mf), in place and out of place, N = 2 .. 65536; the
    values are identical bit for bit, lazy outputs included."""
    import ctypes as C
    if not ho.lib.ho_has_avx512():
        pytest.skip("host CPU has no AVX-512 F/DQ")
    rng = np.random.default_rng(3)
    for n, bits in ((2, 20), (8, 30), (16, 54), (64, 61), (1024, 45), (8192, 54), (65536, 54),
                    (65536, 61)):
        q = ho.generate_primes(1, bits, True, n)[0]
        plan = ho.lib.ho_ntt_create(n, q, 0)
        P = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint64))
        for fwd, in_mf, out_mf in ((1, 1, 1), (1, 4, 4), (1, 2, 1), (1, 4, 1), (0, 1, 1),
                                   (0, 2, 2), (0, 2, 1), (0, 1, 2)):
            x = rng.integers(0, in_mf * q, 2 * n, dtype=np.uint64)
            x[0] = in_mf * q - 1
            a, b = np.empty_like(x), np.empty_like(x)
            scalar = ho.lib.ho_ntt_forward_batch if fwd else ho.lib.ho_ntt_inverse_batch
            simd = ho.lib.ho_ntt_forward_batch_avx512 if fwd else ho.lib.ho_ntt_inverse_batch_avx512
            scalar(plan, P(a), P(x), 2, in_mf, out_mf)
            simd(plan, P(b), P(x), 2, in_mf, out_mf)
            assert np.array_equal(a, b), (n, bits, fwd, in_mf, out_mf)
            y = x.copy()
            simd(plan, P(y), P(y), 2, in_mf, out_mf)  # in place
            assert np.array_equal(a, y)
        ho.lib.ho_ntt_destroy(plan)
