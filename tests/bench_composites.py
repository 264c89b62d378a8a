"""Composite callers on the device: DyadicMultiply and KeySwitch timings
at a CKKS-like shape, next to the oracle (scalar CPU restatement) on one host thread.
Lives under tests/ (not collected by pytest) because it uses the oracle as its checker:
`python tests/bench_composites.py` on a GPU box."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402
from oracle import hexl_oracle as ho  # noqa: E402  (checker / CPU timing only)

rng = np.random.default_rng(1)


def gpu_time(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


# ---- DyadicMultiply: n = 32768, 16 moduli
n, k = 32768, 16
moduli = [int(q) for q in ho.generate_primes(k, 54, True, n)]
x = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli] * 2)
y = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli] * 2)
dx, dy = hx.from_numpy(x), hx.from_numpy(y)
out = hx.from_numpy(np.zeros(3 * n * k, dtype=np.uint64))
t = gpu_time(lambda: hx.DyadicMultiply(out, dx, dy, n, moduli))
t0 = time.perf_counter()
want = ho.dyadic_multiply(x, y, n, moduli)
tc = time.perf_counter() - t0
assert np.array_equal(hx.to_numpy(out), want)
print(f"DyadicMultiply n={n} x {k} moduli: GPU {t * 1e6:8.1f} us "
      f"({56.0 * n * k / t / 1e9:6.0f} GB/s at 56 B/coefficient), oracle 1 thread {tc * 1e3:7.2f} ms")

# ---- KeySwitch: CKKS-like shapes (decomposition moduli + special prime, 2 key components)
def key_switch_shape(n, D, bits):
    K, C = D + 1, 2
    moduli = [int(q) for q in ho.generate_primes(K, bits, True, n)]
    target = np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64) for j in range(D)])
    keys = [np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                            for _ in range(C) for i in range(K)]) for _ in range(D)]
    msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
    result = np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                             for _ in range(C) for i in range(D)])
    d_keys = [hx.from_numpy(kk) for kk in keys]
    d_t = hx.from_numpy(target)
    d_r = hx.from_numpy(result)
    hx.KeySwitch(d_r, d_t, n, D, K, D + 1, C, moduli, d_keys, msf)
    t0 = time.perf_counter()
    want = ho.key_switch(result, target, n, D, K, D + 1, C, moduli, keys, msf)
    tc = time.perf_counter() - t0
    assert np.array_equal(hx.to_numpy(d_r), want)
    t = gpu_time(lambda: hx.KeySwitch(d_r, d_t, n, D, K, D + 1, C, moduli, d_keys, msf), reps=20)
    print(f"KeySwitch n={n}, {D} decomposition moduli of {bits + 1} bits, {C} key components: "
          f"GPU {t * 1e6:8.1f} us per call, oracle 1 thread {tc * 1e3:7.2f} ms", flush=True)
    # many targets per call
    for T in (4, 16, 64, 256):
        d_tt = hx.from_numpy(np.tile(target, T))
        d_rr = hx.from_numpy(np.tile(result, T))
        hx.KeySwitchBatch(d_rr, d_tt, T, n, D, K, D + 1, C, moduli, d_keys, msf)
        got = hx.to_numpy(d_rr).reshape(T, -1)
        assert np.array_equal(got[0], want) and np.array_equal(got[T - 1], want)
        t = gpu_time(lambda: hx.KeySwitchBatch(d_rr, d_tt, T, n, D, K, D + 1, C, moduli, d_keys, msf),
                     reps=10)
        print(f"  KeySwitchBatch {T:4d} targets: GPU {t * 1e6:9.1f} us per call = "
              f"{t * 1e6 / T:7.1f} us per target", flush=True)


key_switch_shape(16384, 7, 54)
key_switch_shape(8192, 4, 54)
key_switch_shape(8192, 4, 48)
key_switch_shape(32768, 15, 54)
