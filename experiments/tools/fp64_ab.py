"""A/B of the Fp64 arithmetic policy against the integer Lazy policy (q < 2^50):
BASELINE configs[1]'s 50-bit prime at the headline shape (N = 65536 x 4096) and at its
own shape (N = 4096 x 256), per-kernel HIP-event times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402


def kernels(ntt, x, steps):
    for _ in range(3):
        ntt.ComputeForward(x, x, 1, 1)
        ntt.ComputeInverse(x, x, 1, 1)
    torch.cuda.synchronize()
    hx.profile_start(8 * steps + 8)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        ntt.ComputeForward(x, x, 1, 1)
        ntt.ComputeInverse(x, x, 1, 1)
    e1.record()
    torch.cuda.synchronize()
    rec = hx.profile_stop()
    agg = {}
    for name, ms in rec:
        agg.setdefault(name, []).append(ms)
    return e0.elapsed_time(e1) / steps, {k: sum(v) / len(v) for k, v in agg.items()}


BITS = int(sys.argv[1]) if len(sys.argv) > 1 else 49   # 28: Small (fp64 = 0) against Fp64 (fp64 = 2)
ON = 2 if BITS < 30 else 1
for n, batch, steps in ((65536, 4096, 10), (4096, 256, 200)):
    q = hx.GeneratePrimes(1, BITS, True, n)[0]
    x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
    hx.fill_splitmix(x, n, batch, 1, q)
    ref = x[:2].clone()
    for fp in (0, ON, 0, ON):
        hx.set_tuning("fp64", fp)
        ntt = hx.NTT(n, q)
        ms, k = kernels(ntt, x, steps)
        assert torch.equal(ref, x[:2])
        print("N=%d batch=%d q=%d %-5s %8.4f ms/step  %s" % (
            n, batch, q, "fp64" if fp else ("small" if BITS < 30 else "lazy"), ms,
            "  ".join("%s=%.4f" % (a.replace("ntt_", ""), b) for a, b in sorted(k.items()))), flush=True)
    hx.set_tuning("fp64", 1)
