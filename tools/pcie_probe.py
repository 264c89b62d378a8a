"""PCIe facts behind the host-buffer path: one-direction and bidirectional copy rates from
pinned memory, from pageable memory, and what pinning the caller's pages costs."""
import ctypes as C
import time

import numpy as np
import torch

MB = 256
n = MB * (1 << 20) // 8
pin_a = torch.empty(n, dtype=torch.int64).pin_memory()
pin_b = torch.empty(n, dtype=torch.int64).pin_memory()
page_a = torch.empty(n, dtype=torch.int64)
dev_a = torch.empty(n, dtype=torch.int64, device="cuda")
dev_b = torch.empty(n, dtype=torch.int64, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def h2d():
    with torch.cuda.stream(s1):
        dev_a.copy_(pin_a, non_blocking=True)


def d2h():
    with torch.cuda.stream(s2):
        pin_b.copy_(dev_b, non_blocking=True)


def both():
    h2d()
    d2h()


g = MB / 1024
print("H2D pinned      %.1f GB/s" % (g / timed(h2d)))
print("D2H pinned      %.1f GB/s" % (g / timed(d2h)))
print("both directions %.1f GB/s aggregate" % (2 * g / timed(both)))
print("H2D pageable    %.1f GB/s" % (g / timed(lambda: dev_a.copy_(page_a))))
hip = C.CDLL("libamdhip64.so")
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
buf = np.ones(n, dtype=np.int64)
t0 = time.perf_counter()
rc = hip.hipHostRegister(buf.ctypes.data, buf.nbytes, 0)
t1 = time.perf_counter()
rc2 = hip.hipHostUnregister(buf.ctypes.data)
t2 = time.perf_counter()
print("hipHostRegister of %d MiB: rc=%d %.2f ms, unregister rc=%d %.2f ms" % (MB, rc, (t1 - t0) * 1e3, rc2, (t2 - t1) * 1e3))
