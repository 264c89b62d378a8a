"""Single-call latency of one forward transform (one polynomial per call: what an unmodified
intel::hexl caller does) by kind of buffer: ordinary (pageable) host memory -> staged through the
device (H2D, kernel, D2H); pinned, device-mapped host memory (hexl_amd_host_alloc, the
intel::hexl DeviceMappedAllocator) -> the kernel runs straight on it; device memory -> launch
only (+ a synchronisation, to compare like with like).  N = 4096 / 8192 / 16384 (one kernel) and
32768 / 65536 (two passes: the mapped operand is read in place, the result copied back)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

REPS = 300
SIZES = ((4096, 49), (4096, 54), (8192, 54), (16384, 54), (32768, 54), (65536, 54), (131072, 54))
for N, bits in SIZES:
    q = hx.GeneratePrimes(1, bits, True, N)[0]
    ntt = hx.NTT(N, q)
    src = np.random.default_rng(1).integers(0, q, N, dtype=np.uint64)
    want = None
    row = []
    # 1. pageable host memory
    a, b = src.copy(), np.zeros(N, dtype=np.uint64)
    pa, pb = a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)
    for _ in range(20):
        hx.lib.hexl_amd_ntt_forward_host(ntt._h, pb, pa, 1, 1, 1)
    t0 = time.perf_counter()
    for _ in range(REPS):
        hx.lib.hexl_amd_ntt_forward_host(ntt._h, pb, pa, 1, 1, 1)
    row.append((time.perf_counter() - t0) / REPS * 1e6)
    want = b.copy()
    # 2. pinned, device-mapped host memory
    pm = C.c_void_p()
    assert hx.lib.hexl_amd_host_alloc(C.byref(pm), 2 * N * 8) == 0
    assert hx.lib.hexl_amd_pointer_kind(pm) == 2
    m = np.ctypeslib.as_array(C.cast(pm, C.POINTER(C.c_uint64)), shape=(2 * N,))
    m[:N] = src
    po = C.c_void_p(pm.value + N * 8)
    for _ in range(20):
        hx.lib.hexl_amd_ntt_forward_host(ntt._h, po, pm, 1, 1, 1)
    t0 = time.perf_counter()
    for _ in range(REPS):
        hx.lib.hexl_amd_ntt_forward_host(ntt._h, po, pm, 1, 1, 1)
    row.append((time.perf_counter() - t0) / REPS * 1e6)
    assert np.array_equal(m[N:], want)
    hx.lib.hexl_amd_host_free(pm)
    # 3. device memory (launch + synchronise per call)
    d = hx.from_numpy(src)
    o = torch.empty_like(d)
    for _ in range(20):
        ntt.ComputeForward(o, d, 1, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        ntt.ComputeForward(o, d, 1, 1)
        torch.cuda.synchronize()
    row.append((time.perf_counter() - t0) / REPS * 1e6)
    assert np.array_equal(hx.to_numpy(o), want)
    # 4. device memory, launches back to back (no per-call synchronisation)
    t0 = time.perf_counter()
    for _ in range(REPS):
        ntt.ComputeForward(o, d, 1, 1)
    torch.cuda.synchronize()
    row.append((time.perf_counter() - t0) / REPS * 1e6)
    print(f"N={N:6d} q~2^{bits + 1}: pageable host {row[0]:7.1f} us | device-mapped host {row[1]:7.1f} us | "
          f"device + sync {row[2]:7.1f} us | device, queued {row[3]:7.1f} us   per call")
