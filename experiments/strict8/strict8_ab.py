"""A/B of Strict8 against Strict for a prime just below 2^61 (host-marked
subtraction extended to the 64-bit policy): forward and inverse passes over a 1 GiB batch, kernel
times from the library's launch profiler, interleaved A B A B on the same box.  Kill criterion:
the forward transform at N = 2^17 must drop by at least 5 %."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

for n in (4096, 16384, 65536, 1 << 17, 1 << 19):
    q = hx.GeneratePrimes(1, 60, False, n)[0]  # walking down from 2^61
    assert (1 << 60) + (1 << 28) <= q < (1 << 61)
    batch = (1 << 30) // (8 * n)
    x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
    rows, fwd = [], {}
    for name, on in (("Strict8", 1), ("Strict", 0), ("Strict8", 1), ("Strict", 0)):
        hx.set_tuning("strict8", on)
        ntt = hx.NTT(n, q)
        hx.fill_splitmix(x, n, batch, 1, q)
        for _ in range(30):
            ntt.ComputeForward(x, x, 1, 1)
            ntt.ComputeInverse(x, x, 1, 1)
        torch.cuda.synchronize()
        hx.profile_start(512)
        for _ in range(15):
            ntt.ComputeForward(x, x, 1, 1)
        rec = hx.profile_stop()
        f = sum(v for _, v in rec) / 15
        per = {}
        for k, v in rec:
            per[k] = per.get(k, 0.0) + v / 15
        hx.profile_start(512)
        for _ in range(15):
            ntt.ComputeInverse(x, x, 1, 1)
        i = sum(v for _, v in hx.profile_stop()) / 15
        fwd.setdefault(name, []).append(f)
        rows.append(f"{name} {f:.3f}/{i:.3f} (" + ", ".join(f"{k.replace('ntt_fwd_', '')} {v:.3f}" for k, v in per.items()) + ")")
    hx.set_tuning("strict8", 1)
    a, b = min(fwd["Strict8"]), min(fwd["Strict"])
    print(f"N={n} q=2^61-{(1 << 61) - q} batch={batch}: fwd/inv ms  " + " | ".join(rows)
          + f"  => forward {100 * (a / b - 1):+.1f} %", flush=True)
