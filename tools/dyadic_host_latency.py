"""Per-call time of hexl_amd_dyadic_multiply_host (what intel::hexl::DyadicMultiply binds for caller memory) on ordinary
host memory by size, with the bounce route on and off ("host_bounce_kb")."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

rng = np.random.default_rng(3)
p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731


def per_call(fn, reps=300):
    for _ in range(30):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


for n, k in ((1024, 2), (4096, 1), (4096, 2), (8192, 1), (8192, 2), (16384, 3)):
    moduli = [int(q) for q in hx.GeneratePrimes(k, 54, True, n)]
    x = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli] * 2)
    y = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli] * 2)
    out = np.zeros(3 * n * k, dtype=np.uint64)
    mod = (C.c_uint64 * k)(*moduli)
    row, ref = [], None
    for kb in (0, 512, 0, 512):
        hx.set_tuning("host_bounce_kb", kb)
        row.append(f"{per_call(lambda: hx.lib.hexl_amd_dyadic_multiply_host(p(out), p(x), p(y), n, mod, k)):.1f}")
        if ref is None:
            ref = out.copy()
        assert np.array_equal(out, ref)
    print(f"DyadicMultiply host n={n} x {k} moduli ({7 * n * k * 8 >> 10} KiB in + out): staged {row[0]} {row[2]} | "
          f"bounce limit 512 KiB {row[1]} {row[3]} us per call", flush=True)
hx.set_tuning("host_bounce_kb", 512)
