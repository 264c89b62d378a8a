"""A/B of the Harvey60 arithmetic policy against Strict for 2^56 <= q < 2^60 at the
headline shape (argv[1] = GeneratePrimes bit size, default 59: a 60-bit prime)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 59
for N, B in ((65536, 4096), (4096, 65536), (16384, 16384), (131072, 1024)):
    q = hx.GeneratePrimes(1, bits, False, N)[0]
    plans = {}
    for h in (0, 1):
        hx.set_tuning("h60", h)
        plans[h] = hx.NTT(N, q)
    hx.set_tuning("h60", 1)
    x = torch.empty((B, N), dtype=torch.int64, device="cuda")
    hx.fill_splitmix(x, N, B, 1, q)
    ref = x[:2].clone()
    hx.profile_start(4096)
    for rep in range(2):
        for h in (0, 1):
            ntt = plans[h]
            for _ in range(5):
                ntt.ComputeForward(x, x, 1, 1)
                ntt.ComputeInverse(x, x, 1, 1)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ntt.ComputeForward(x, x, 1, 1)
                ntt.ComputeInverse(x, x, 1, 1)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            print("N=%6d x %5d, q=%d: %-8s %7.3f ms/step  %6.2f M NTT/s" % (
                N, B, q, "harvey60" if h else "strict", ms, 2 * B / ms / 1e3), flush=True)
    hx.profile_stop()
    assert torch.equal(ref, x[:2])
    del x
