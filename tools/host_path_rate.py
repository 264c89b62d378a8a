"""PCIe-inclusive rate of the host-buffer entry points (what an unmodified HEXL caller with
host vectors sees): hexl_amd_ntt_forward_host / inverse_host on pageable numpy buffers."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

N, q = 65536, 18014398510661633
ntt = hx.NTT(N, q)
for batch in (1, 16, 64, 256, 1024):
    x = np.random.default_rng(1).integers(0, q, (batch, N), dtype=np.uint64)
    ref = x.copy()
    p = x.ctypes.data_as(C.c_void_p)
    for _ in range(2):
        hx.lib.hexl_amd_ntt_forward_host(ntt._h, p, p, batch, 1, 1)
        hx.lib.hexl_amd_ntt_inverse_host(ntt._h, p, p, batch, 1, 1)
    reps = max(2, 64 // batch)
    t0 = time.perf_counter()
    for _ in range(reps):
        hx.lib.hexl_amd_ntt_forward_host(ntt._h, p, p, batch, 1, 1)
        hx.lib.hexl_amd_ntt_inverse_host(ntt._h, p, p, batch, 1, 1)
    dt = (time.perf_counter() - t0) / (2 * reps)
    assert np.array_equal(x, ref)
    print(f"host path, N={N}, batch {batch:4d}: {dt * 1e3:8.3f} ms per call, "
          f"{batch / dt / 1e3:7.1f} k NTT/s, {2 * batch * N * 8 / dt / 1e9:6.1f} GB/s over PCIe (in+out)")
