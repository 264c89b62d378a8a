"""A/B of the arithmetic policies a modulus in [2^56, 2^59) can take: the bounded members of the
Lazy family (default) against Harvey60 ("lazy_family" = 0) against Strict ("h60" = 0 too), the
fwd+inv step over `batch` polynomials, per-kernel HIP-event times.
python tools/lazy_family_ab.py [N] [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
x = torch.empty((B, N), dtype=torch.int64, device="cuda")
for bits, small in ((56, True), (57, False), (58, True), (58, False)):
    q = hx.GeneratePrimes(1, bits, small, N)[0]
    rows = []
    for name, fam, h60 in (("lazy family", 1, 1), ("Harvey60", 0, 1), ("Strict", 0, 0)):
        hx.set_tuning("lazy_family", fam)
        hx.set_tuning("h60", h60)
        ntt = hx.NTT(N, q)
        hx.fill_splitmix(x, N, B, 1, q)
        ref = x[:1].clone()

        def step():
            ntt.ComputeForward(x, x, 1, 1)
            ntt.ComputeInverse(x, x, 1, 1)
        for _ in range(25):
            step()
        torch.cuda.synchronize()
        # the step (HIP events around 30 steps) and its kernels (the library's launch profiler:
        # an event pair per launch) in the SAME 30 steps, so that the kernel columns sum to the
        # step up to the launch gaps (round 4 profiled 5 separate steps after the timed ones: a
        # different pass of the box, and its columns did not add up)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        hx.profile_start(8 * 30 + 16)
        e0.record()
        for _ in range(30):
            step()
        e1.record()
        torch.cuda.synchronize()
        records = hx.profile_stop()
        assert torch.equal(ref, x[:1])
        agg = {}
        for k, v in records:
            agg.setdefault(k.replace("ntt_", "").replace("_pass", "").replace("_bottom", ""), []).append(v)
        kern = {k: round(sum(v) / 30, 3) for k, v in agg.items()}
        kern["sum"] = round(sum(sum(v) for v in agg.values()) / 30, 3)
        rows.append((name, e0.elapsed_time(e1) / 30, kern))
    hx.set_tuning("lazy_family", 1)
    hx.set_tuning("h60", 1)
    print(f"N={N} batch={B} q={q} ({q.bit_length()} bits, 2^63/q = {(1 << 63) // q}):")
    for name, ms, kern in rows:
        print(f"   {name:12s} {ms:7.3f} ms per step  {2 * B / ms / 1e3:7.3f} M NTT/s  {kern}", flush=True)
