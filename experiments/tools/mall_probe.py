"""Is a working set that fits the 256 MiB Infinity Cache served faster than HBM, and
does a multi-stream chunked transform (each chunk's second pass following its first
while the chunk is cache resident) exploit it?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


print("== in-place add / copy bandwidth against footprint", flush=True)
for mib in (16, 32, 64, 128, 192, 256, 512, 2048):
    n = mib * (1 << 20) // 8
    a = torch.zeros(n, dtype=torch.int64, device="cuda")
    b = torch.zeros(n, dtype=torch.int64, device="cuda")
    reps = max(20, 4096 // mib)
    t = timed(lambda: a.add_(1), reps)
    t2 = timed(lambda: b.copy_(a), reps)
    print("  %5d MiB: in-place add %.2f TB/s   copy (2x footprint) %.2f TB/s" % (
        mib, 2 * n * 8 / t / 1e12, 2 * n * 8 / t2 / 1e12), flush=True)
    del a, b

hx.set_tuning("plan", hx.PLAN_SPLIT)
N, B = 65536, 4096
q = 18014398510661633
ntt = hx.NTT(N, q)
x = torch.empty((B, N), dtype=torch.int64, device="cuda")
hx.fill_splitmix(x, N, B, 1, q)
ref = x[:2].clone()
streams = [torch.cuda.Stream() for _ in range(4)]


def step(chunk, ns):
    if ns == 0:
        for c in range(0, B, chunk):
            v = x[c:c + chunk]
            ntt.ComputeForward(v, v, 1, 1)
        for c in range(0, B, chunk):
            v = x[c:c + chunk]
            ntt.ComputeInverse(v, v, 1, 1)
        return
    for fwd in (True, False):
        for i, c in enumerate(range(0, B, chunk)):
            v = x[c:c + chunk]
            with torch.cuda.stream(streams[i % ns]):
                (ntt.ComputeForward if fwd else ntt.ComputeInverse)(v, v, 1, 1)
        # the inverse of a chunk follows its forward on the same stream; nothing else to order


print("== chunked transform, ms per fwd+inv step over 4096 polynomials", flush=True)
for _ in range(10):
    step(B, 0)
for chunk in (4096, 1024, 512, 256, 128, 64):
    row = []
    for ns in (0, 1, 2, 3, 4):
        if chunk == 4096 and ns > 0:
            continue
        torch.cuda.synchronize()
        t = min(timed(lambda: step(chunk, ns), 5) for _ in range(3))
        row.append("%d streams %.3f" % (ns, t * 1e3))
    print("  chunk %4d (%4d MiB): " % (chunk, chunk // 2) + "  ".join(row), flush=True)
torch.cuda.synchronize()
assert torch.equal(ref, x[:2])
