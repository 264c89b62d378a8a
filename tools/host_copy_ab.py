"""What the pinned-slot copies of the host paths cost against handing pageable memory to the runtime
("host_direct_copy" 0 / 1): hexl_amd_ntt_forward_host on ordinary numpy memory, in place, wall time per call."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

print("| N x batch | bytes | pinned slots (default) us | direct us | GB/s in+out (slots / direct) |")
print("|---|---|---|---|---|")
THREADS = int(os.environ.get("HOST_COPY_THREADS", "0"))
if THREADS:
    hx.set_tuning("host_copy_threads", THREADS)
for n, batch in ((65536, 1), (131072, 1), (65536, 8), (65536, 16), (65536, 32), (65536, 128), (16384, 512)):
    q = hx.GeneratePrimes(1, 54, True, n)[0]
    ntt = hx.NTT(n, q)
    x = np.random.default_rng(1).integers(0, q, (batch, n), dtype=np.uint64)
    p = x.ctypes.data_as(C.c_void_p)
    row = []
    for direct in (0, 1, 0, 1):
        hx.set_tuning("host_direct_copy", direct)
        reps = 40 if batch <= 8 else 10
        for _ in range(3):
            assert hx.lib.hexl_amd_ntt_forward_host(ntt._h, p, p, batch, 4, 4) == 0
        t0 = time.perf_counter()
        for _ in range(reps):
            hx.lib.hexl_amd_ntt_forward_host(ntt._h, p, p, batch, 4, 4)
        row.append((time.perf_counter() - t0) / reps * 1e6)
    hx.set_tuning("host_direct_copy", 0)
    a, b = min(row[0], row[2]), min(row[1], row[3])
    print(f"| {n} x {batch} | {x.nbytes >> 10} KiB | {a:.0f} | {b:.0f} | {2 * x.nbytes / a / 1e3:.1f} / {2 * x.nbytes / b / 1e3:.1f} |", flush=True)
torch.cuda.synchronize()
