"""Does running the two passes of the transform chunk by chunk (chunk sized to the
256 MiB Infinity Cache or the 32 MiB of L2) beat whole-batch passes?"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

hx.set_tuning("plan", hx.PLAN_SPLIT)
N, B = 65536, 4096
q = 18014398510661633
ntt = hx.NTT(N, q)
x = torch.empty((B, N), dtype=torch.int64, device="cuda")
hx.fill_splitmix(x, N, B, 1, q)
ref = x[:2].clone()


def step(chunk):
    for c in range(0, B, chunk):
        v = x[c:c + chunk]
        ntt.ComputeForward(v, v, 1, 1)
    for c in range(0, B, chunk):
        v = x[c:c + chunk]
        ntt.ComputeInverse(v, v, 1, 1)


for _ in range(10):
    step(B)
for chunk in (4096, 1024, 512, 256, 128, 64, 32):
    for _ in range(2):
        step(chunk)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(5):
            step(chunk)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 5)
    print("chunk %4d polys (%4d MiB): %.3f ms/step" % (chunk, chunk // 2, best * 1e3), flush=True)
assert torch.equal(ref, x[:2])
