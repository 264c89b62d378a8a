"""Latency A/B for the smallest batches at N = 8192: the one-kernel plan (64 KiB LDS tile, one workgroup of 1024
threads per polynomial) against the two-pass plan (2 strided stages + the 11-stage tile pass on four workgroups per
polynomial; "tile13" = 0), as WALL time per call of back-to-back dependent calls on one stream -- what a caller with
one polynomial at a time sees -- and the one-polynomial host call."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

n = 8192


def wall(fn, reps=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for bits in (54, 49):
    q = hx.GeneratePrimes(1, bits, True, n)[0]
    ntt = hx.NTT(n, q)
    print(f"## {bits + 1}-bit prime")
    print("| batch | two-pass fwd / inv us | one-kernel fwd / inv us |")
    print("|---|---|---|")
    for batch in (1, 2, 4, 8, 16, 32, 64, 128, 256):
        x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
        hx.fill_splitmix(x, n, batch, 1, q)
        cells = []
        for rep in range(2):
            for t13 in (0, 2):
                hx.set_tuning("tile13", t13)
                f = wall(lambda: ntt.ComputeForward(x, x, 1, 1))
                i = wall(lambda: ntt.ComputeInverse(x, x, 1, 1))
                cells.append(f"{f:.1f} / {i:.1f}")
        print(f"| {batch} | {cells[0]} ; {cells[2]} | {cells[1]} ; {cells[3]} |", flush=True)
    a = np.random.default_rng(1).integers(0, q, n, dtype=np.uint64)
    b = np.zeros_like(a)
    pa, pb = a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)
    for t13 in (0, 2, 0, 2):
        hx.set_tuning("tile13", t13)
        t = wall(lambda: hx.lib.hexl_amd_ntt_forward_host(ntt._h, pb, pa, 1, 1, 1))
        print(f"host call, one polynomial, tile13={t13}: {t:.1f} us")
hx.set_tuning("tile13", 2)
