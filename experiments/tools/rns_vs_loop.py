"""configs[3] on one GPU (8 primes x 4096 polynomials of N = 65536, 16 GiB): the multi-modulus
entry point (one launch sequence) against a loop of single-modulus calls over the same buffer,
and against the same loop over ONE prime's 2 GiB slice (the footprint of the headline bench)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

n, b = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
primes = [18014398510661633, 18014398512365569, 18014398514200577, 18014398514987009,
          18014398515511297, 18014398516559873, 18014398521016321, 18014398524424193]
plans = [hx.NTT(n, p) for p in primes]
x = torch.empty((len(primes), b, n), dtype=torch.int64, device="cuda")
for k, p in enumerate(primes):
    hx.fill_splitmix(x[k], n, b, 1 + k * b, p)


def timed(fn, reps=5):
    for _ in range(8):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def rns():
    hx.ComputeForwardRNS(plans, x, x, 1, 1)
    hx.ComputeInverseRNS(plans, x, x, 1, 1)


def loop():
    for k, pl in enumerate(plans):
        pl.ComputeForward(x[k], x[k], 1, 1)
    for k, pl in enumerate(plans):
        pl.ComputeInverse(x[k], x[k], 1, 1)


def one_slice():
    for _ in range(8):
        plans[0].ComputeForward(x[0], x[0], 1, 1)
    for _ in range(8):
        plans[0].ComputeInverse(x[0], x[0], 1, 1)


for rep in range(2):
    print("multi-modulus entry point : %.2f ms per step over 8 x %d polynomials" % (timed(rns), b), flush=True)
    print("loop of single-modulus calls: %.2f ms" % timed(loop), flush=True)
    print("the same 16 launches on one 2 GiB slice: %.2f ms" % timed(one_slice), flush=True)

for name, fn in (("multi", rns), ("loop", loop)):
    hx.profile_start(512)
    fn()
    torch.cuda.synchronize()
    rec = hx.profile_stop()
    agg = {}
    for k, ms in rec:
        agg.setdefault(k, []).append(ms)
    print(name, {k: "%d x %.3f ms" % (len(v), sum(v) / len(v)) for k, v in agg.items()}, flush=True)
