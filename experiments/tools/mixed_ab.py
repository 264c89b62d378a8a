"""A/B at the headline shape: the mixed plan (4 + 12 stages, both passes' workgroups in one
launch, chunk pipeline) against the default two-launch plan (5 + 11)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

N, B = 65536, 4096
bits = int(sys.argv[1]) if len(sys.argv) > 1 else 54
q = hx.GeneratePrimes(1, bits, True, N)[0] if bits != 54 else 18014398510661633
ntt = hx.NTT(N, q)
x = torch.empty((B, N), dtype=torch.int64, device="cuda")
hx.fill_splitmix(x, N, B, 1, q)
ref = x[:2].clone()


def step():
    ntt.ComputeForward(x, x, 1, 1)
    ntt.ComputeInverse(x, x, 1, 1)


def timed(steps=20):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            step()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / steps)
    return best


for _ in range(10):
    step()
for rep in range(2):
    hx.set_tuning("plan", hx.PLAN_SPLIT)
    print("split (5 + 11, two launches): %.3f ms/step" % timed(), flush=True)
    for chunk in (2048, 1024, 512, 256, 128, 64):
        hx.set_tuning("plan", hx.PLAN_MIXED)
        hx.set_tuning("mixed_chunk", chunk)
        print("mixed, chunk %4d: %.3f ms/step" % (chunk, timed()), flush=True)
hx.set_tuning("plan", hx.PLAN_SPLIT)
torch.cuda.synchronize()
assert torch.equal(ref, x[:2])
