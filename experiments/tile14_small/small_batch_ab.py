"""Latency A/B for the smallest batches at N = 16384 ("tile14_small"): the one-kernel plan (a
128 KiB LDS tile, one workgroup per polynomial) against the two-pass plan, as WALL time per call of
back-to-back dependent calls on one stream (launch gaps included -- the quantity a caller with one
ciphertext at a time sees), for plain transforms and for KeySwitch one target per call (launch by
launch and replayed from its captured graph).  Also the one-polynomial host call."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

n = 16384
q = hx.GeneratePrimes(1, 54, True, n)[0]
side = torch.cuda.Stream()


def wall(fn, reps=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


print("| batch | two-pass fwd / inv us | one-kernel fwd / inv us |")
print("|---|---|---|")
for batch in (1, 2, 7, 14, 49, 64, 100, 150):
    x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
    hx.fill_splitmix(x, n, batch, 1, q)
    cells = []
    for small in (0, 191):
        hx.set_tuning("tile14_small", small)
        ntt = hx.NTT(n, q)
        f = wall(lambda: ntt.ComputeForward(x, x, 1, 1))
        i = wall(lambda: ntt.ComputeInverse(x, x, 1, 1))
        cells.append(f"{f:.1f} / {i:.1f}")
    print(f"| {batch} | {cells[0]} | {cells[1]} |", flush=True)

# one polynomial in ordinary host memory through the *_host entry point
a = np.random.default_rng(1).integers(0, q, n, dtype=np.uint64)
b = np.zeros_like(a)
pa, pb = a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)
for small in (0, 191):
    hx.set_tuning("tile14_small", small)
    ntt = hx.NTT(n, q)
    t = wall(lambda: hx.lib.hexl_amd_ntt_forward_host(ntt._h, pb, pa, 1, 1, 1))
    print(f"host call, one polynomial, tile14_small={small}: {t:.1f} us")

# KeySwitch, one target per call
D, Cc = 7, 2
K = D + 1
rng = np.random.default_rng(1)
moduli = hx.GeneratePrimes(K, 54, True, n)
keys = [hx.from_numpy(np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                                      for _ in range(Cc) for i in range(K)])) for _ in range(D)]
msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
d_t = hx.from_numpy(np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64) for j in range(D)]))
d_r = hx.from_numpy(np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                                    for _ in range(Cc) for i in range(D)]))


def ks():
    with torch.cuda.stream(side):
        hx.KeySwitch(d_r, d_t, n, D, K, D + 1, Cc, moduli, keys, msf)


for small in (0, 191):
    hx.set_tuning("tile14_small", small)
    for graph in (0, 1):
        hx.set_tuning("ks_graph", graph)
        hx.lib.hexl_amd_release_stream_workspaces(side.cuda_stream)
        r0 = hx.get_counter("ks_graph_replays")
        t = wall(ks, 100)
        print(f"KeySwitch one target, n={n} D={D}: tile14_small={small} ks_graph={graph}: {t:.1f} us per call "
              f"({hx.get_counter('ks_graph_replays') - r0} replays)", flush=True)
hx.set_tuning("tile14_small", 0)
hx.set_tuning("ks_graph", 1)
