"""A/B of the two members of the Fp64 family for moduli below 2^47: long runs (default) against
the short-run Fp64 ("fp64_long" = 0): forward and inverse passes over a 1 GiB batch, kernel times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

for n, bits in ((4096, 36), (8192, 43), (16384, 44), (65536, 44), (4096, 46)):
    q = hx.GeneratePrimes(1, bits, True, n)[0]
    batch = (1 << 30) // (8 * n)
    x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
    rows = []
    for name, lr in (("Fp64L", 1), ("Fp64", 0), ("Fp64L", 1), ("Fp64", 0)):
        hx.set_tuning("fp64_long", lr)
        ntt = hx.NTT(n, q)
        hx.fill_splitmix(x, n, batch, 1, q)
        for _ in range(30):
            ntt.ComputeForward(x, x, 1, 1)
            ntt.ComputeInverse(x, x, 1, 1)
        torch.cuda.synchronize()
        hx.profile_start(512)
        for _ in range(15):
            ntt.ComputeForward(x, x, 1, 1)
        f = sum(v for _, v in hx.profile_stop()) / 15
        hx.profile_start(512)
        for _ in range(15):
            ntt.ComputeInverse(x, x, 1, 1)
        i = sum(v for _, v in hx.profile_stop()) / 15
        rows.append(f"{name} {f:.3f}/{i:.3f}")
    hx.set_tuning("fp64_long", 1)
    print(f"N={n} q~2^{bits} batch={batch}: fwd/inv ms  " + " | ".join(rows), flush=True)
