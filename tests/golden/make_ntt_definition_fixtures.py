"""Generates tests/golden/ntt_definition_fixtures.json: known answers of the negacyclic
NTT at the benchmark's sizes, computed from the transform's DEFINITION in Python big
integers -- independently of the oracle and of the HIP path:

    forward:  out[i] = sum_j a[j] * w^((2*bitrev(i)+1)*j)  mod q
    inverse:  out[j] = N^-1 * sum_i X[i] * w^(-(2*bitrev(i)+1)*j)  mod q

(w = the minimal primitive 2N-th root of unity mod q, what intel::hexl::NTT(N, q) uses --
hexl/ntt/ntt-internal.cpp:60-72; input and output orders as the
reference: forward natural -> bit-reversed, inverse bit-reversed -> natural).

Everything here (splitmix64 input stream, root search, evaluation) is pure Python; the
oracle is only consulted at the end, to (a) assert that it agrees on every sampled entry
and (b) record the SHA-256 of its full output vector, which the tests then use as the
whole-vector pin for both the oracle and the HIP path.

    python tests/golden/make_ntt_definition_fixtures.py        (~4 min)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

M64 = (1 << 64) - 1
CASES = [  # (N, q, what it pins)
    (4096, 562949954093057, "BASELINE configs[1]: N=4096, 50-bit prime"),
    (65536, 18014398510661633, "BASELINE configs[2] (headline): N=65536, 55-bit prime"),
    (131072, 1152921504616808449, "BASELINE configs[4]: N=131072, 61-bit prime"),
] + [  # the other seven RNS primes of configs[3] (bench.py's per-rank probe: every rank checks the
       # transform of ITS prime on ITS device against these digests before it is timed)
    (65536, q, "BASELINE configs[3]: RNS prime %d of 8, N=65536" % (k + 2))
    for k, q in enumerate([18014398512365569, 18014398514200577, 18014398514987009, 18014398515511297,
                           18014398516559873, 18014398521016321, 18014398524424193])
]
SAMPLES = 64


def splitmix(n, seed, bound):
    """coefficient i of polynomial `seed` = i-th output of splitmix64(seed) mod bound"""
    out, s = [], seed
    for _ in range(n):
        s = (s + 0x9E3779B97F4A7C15) & M64
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
        z ^= z >> 31
        out.append(z % bound)
    return out


def bitrev(x, bits):
    return int(format(x, "0%db" % bits)[::-1], 2)


def minimal_primitive_root(two_n, q):
    """smallest w with w^(two_n/2) == -1 mod q"""
    x = 2
    while True:
        r = pow(x, (q - 1) // two_n, q)
        if pow(r, two_n // 2, q) == q - 1:
            break
        x += 1
    best, cur, r2 = r, r, r * r % q
    for _ in range(two_n // 2 - 1):  # all odd powers of r = all primitive roots
        cur = cur * r2 % q
        if cur < best:
            best = cur
    return best


def sample_indices(n, seed):
    idx = list(range(4)) + [n - 4 + k for k in range(4)] + [n // 2 - 1, n // 2]
    for v in splitmix(4 * SAMPLES, seed, n):
        if len(idx) >= SAMPLES:
            break
        if v not in idx:
            idx.append(v)
    return idx


def main():
    from oracle import hexl_oracle as ho

    fixtures = []
    for n, q, what in CASES:
        bits = n.bit_length() - 1
        w = minimal_primitive_root(2 * n, q)
        a = splitmix(n, 1, q)         # forward input, natural order
        X = splitmix(n, 1001, q)      # inverse input, bit-reversed order
        fwd = []
        for i in sample_indices(n, 7):
            r = pow(w, 2 * bitrev(i, bits) + 1, q)
            acc = 0
            for c in reversed(a):     # Horner
                acc = (acc * r + c) % q
            fwd.append([i, acc])
        w_inv = pow(w, q - 2, q)
        pw = [1] * (2 * n)
        for k in range(1, 2 * n):
            pw[k] = pw[k - 1] * w_inv % q
        expo = [2 * bitrev(i, bits) + 1 for i in range(n)]
        n_inv = pow(n, q - 2, q)
        inv = []
        mask = 2 * n - 1
        for j in sample_indices(n, 8):
            acc = 0
            for i in range(n):
                acc += X[i] * pw[(expo[i] * j) & mask]
            inv.append([j, acc % q * n_inv % q])
        # the oracle must agree on every sampled entry; its digest pins the whole vector
        ont = ho.NTT(n, q)
        assert ont.w == w, (ont.w, w)
        assert [int(v) for v in ho.fill_splitmix(n, 1, q)] == a
        of = ont.forward(np.array(a, dtype=np.uint64), 1, 1)
        oi = ont.inverse(np.array(X, dtype=np.uint64), 1, 1)
        assert all(int(of[i]) == v for i, v in fwd)
        assert all(int(oi[j]) == v for j, v in inv)
        fixtures.append({
            "what": what, "n": n, "q": q, "minimal_root": w,
            "forward": {"input": "splitmix64(seed=1) mod q, natural order", "seed": 1,
                        "samples": fwd,
                        "sha256_le_u64": hashlib.sha256(of.astype("<u8").tobytes()).hexdigest()},
            "inverse": {"input": "splitmix64(seed=1001) mod q, bit-reversed order", "seed": 1001,
                        "samples": inv,
                        "sha256_le_u64": hashlib.sha256(oi.astype("<u8").tobytes()).hexdigest()},
        })
        print("N=%d q=%d: %d forward + %d inverse entries from the definition" % (
            n, q, len(fwd), len(inv)), flush=True)
    out = {
        "_comment": ("Known answers from the big-integer definition of the transform; generated by "
                     "tests/golden/make_ntt_definition_fixtures.py (see its docstring). 'samples' are "
                     "[index, value] pairs; 'sha256_le_u64' is the digest of the full output vector "
                     "(little-endian uint64) whose sampled entries equal the definition."),
        "cases": fixtures,
    }
    with open(os.path.join(HERE, "ntt_definition_fixtures.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
