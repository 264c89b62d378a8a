"""Per-call time of the host-pointer element-wise entry point (what intel::hexl::EltwiseMultMod / EltwiseFMAMod bind
for ordinary vectors) by size, with the completion flag polled or the stream synchronised ("host_poll") and by the
bounce-buffer limit ("host_bounce_kb").  One line per (n, op)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

q = hx.GeneratePrimes(1, 54, True, 4096)[0]
rng = np.random.default_rng(2)


def per_call(fn, reps):
    for _ in range(20):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps * 1e6


for n in (4096, 16384, 65536, 131072, 262144):
    a = rng.integers(0, q, n, dtype=np.uint64)
    b = rng.integers(0, q, n, dtype=np.uint64)
    r = np.zeros(n, dtype=np.uint64)
    pa, pb, pr = (x.ctypes.data_as(C.c_void_p) for x in (a, b, r))
    want = (a.astype(object) * b.astype(object)) % q
    for name, call in (("mult", lambda: hx.lib.hexl_amd_eltwise_host(4, pr, pa, pb, 0, n, q, 1, 1)),
                       ("fma", lambda: hx.lib.hexl_amd_eltwise_host(5, pr, pa, pb, 12345, n, q, 1, 1))):
        row = []
        for kb in (256, 512, 1024, 2048):
            for poll in (0, 1):
                hx.set_tuning("host_bounce_kb", kb)
                hx.set_tuning("host_poll", poll)
                row.append(f"{kb}K/{'poll' if poll else 'sync'} {per_call(call, 300):6.1f}")
        if name == "mult":
            assert (r.astype(object) == want).all()
        print(f"n={n:7d} {name:5s} us per call: " + " | ".join(row), flush=True)
hx.set_tuning("host_bounce_kb", 256)
hx.set_tuning("host_poll", 1)
