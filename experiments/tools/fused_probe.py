"""GPU probe of fused_pass (one-launch transform): bit-exact comparison against the
two-launch plan over N / arithmetic policy / batch / aliasing, then a timing sweep
over the scheduler's knobs at the headline shape.

    python tools/fused_probe.py [check] [time] [timeN]
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

what = set(sys.argv[1:]) or {"check", "time"}


def transform(ntt, x, fwd, plan, inplace, out_mf=1):
    hx.set_tuning("plan", plan)
    src = x.clone()
    dst = src if inplace else torch.empty_like(src)
    (ntt.ComputeForward if fwd else ntt.ComputeInverse)(dst, src, 1, out_mf)
    torch.cuda.synchronize()
    return dst


if "check" in what:
    hx.set_tuning("fused_min_batch", 1)
    bad = 0
    for logn in (15, 16):
        n = 1 << logn
        for bits in (28, 54, 60):
            q = hx.GeneratePrimes(1, bits, True, n)[0]
            ntt = hx.NTT(n, q)
            for batch in (1, 67, 512):
                x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
                hx.fill_splitmix(x, n, batch, 11 + logn, q)
                for fwd in (True, False):
                    for inplace in (True, False):
                        for out_mf in ((1, 4) if fwd else (1, 2)):
                            a = transform(ntt, x, fwd, hx.PLAN_SPLIT, inplace, out_mf)
                            b = transform(ntt, x, fwd, hx.PLAN_FUSED, inplace, out_mf)
                            if out_mf != 1:  # lazy outputs: congruent and in range
                                ok = bool(((a.to(torch.float64) >= 0).all())) and torch.equal(a % q, b % q) \
                                    and int(b.max()) < out_mf * q and int(b.min()) >= 0
                            else:
                                ok = torch.equal(a, b)
                            if not ok:
                                bad += 1
                                print("MISMATCH", logn, bits, batch, fwd, inplace, out_mf,
                                      int((a != b).sum()))
            print(f"N=2^{logn} {bits}-bit: checked", flush=True)
    # headline shape, full batch
    n, batch, q = 65536, 4096, 18014398510661633
    ntt = hx.NTT(n, q)
    x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
    hx.fill_splitmix(x, n, batch, 1, q)
    for fwd in (True, False):
        a = transform(ntt, x, fwd, hx.PLAN_SPLIT, True)
        for rep in range(3):
            b = transform(ntt, x, fwd, hx.PLAN_FUSED, True)
            if not torch.equal(a, b):
                bad += 1
                print("MISMATCH headline", fwd, rep, int((a != b).sum()))
        del a, b
    print("check:", "FAILED" if bad else "all bit-exact", flush=True)
    hx.set_tuning("fused_min_batch", 64)
    del x


def time_step(ntt, x, steps=10, reps=3):
    for _ in range(3):
        ntt.ComputeForward(x, x, 1, 1)
        ntt.ComputeInverse(x, x, 1, 1)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(steps):
            ntt.ComputeForward(x, x, 1, 1)
            ntt.ComputeInverse(x, x, 1, 1)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps)
    return best * 1e3


if "time" in what:
    n, batch, q = 65536, 4096, 18014398510661633
    ntt = hx.NTT(n, q)
    x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
    hx.fill_splitmix(x, n, batch, 1, q)
    ref = x[:2].clone()
    hx.set_tuning("plan", hx.PLAN_SPLIT)
    for _ in range(15):  # clock / page warm-up
        ntt.ComputeForward(x, x, 1, 1)
        ntt.ComputeInverse(x, x, 1, 1)
    print("split: %.3f ms/step" % time_step(ntt, x), flush=True)
    hx.set_tuning("plan", hx.PLAN_FUSED)
    for wg in (0,):
        for window in (2, 4, 5, 6, 8, 12, 24):
            hx.set_tuning("fused_wg_per_cu", wg)
            hx.set_tuning("fused_window", window)
            ms = time_step(ntt, x, steps=6, reps=2)
            print("fused wg/cu=%d window=%2d: %.3f ms/step  %.3f M NTT/s" % (wg, window, ms, 2 * batch / ms / 1e3),
                  flush=True)
    assert torch.equal(ref, x[:2])
    hx.set_tuning("plan", hx.PLAN_SPLIT)
    print("split again: %.3f ms/step" % time_step(ntt, x), flush=True)
    hx.set_tuning("fused_wg_per_cu", 0)
    hx.set_tuning("fused_window", 12)

if "timeN" in what:  # other degrees / policies at batch 4096-equivalent bytes
    for logn, bits in ((15, 54), (16, 28), (16, 60)):
        n = 1 << logn
        batch = (4096 * 65536) >> logn
        q = hx.GeneratePrimes(1, bits, True, n)[0]
        ntt = hx.NTT(n, q)
        x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
        hx.fill_splitmix(x, n, batch, 1, q)
        for plan, name in ((hx.PLAN_SPLIT, "split"), (hx.PLAN_FUSED, "fused")):
            hx.set_tuning("plan", plan)
            print("N=2^%d %d-bit batch=%d %s: %.3f ms/step" % (logn, bits, batch, name, time_step(ntt, x, 6, 2)),
                  flush=True)
        del x
