"""A/B: N = 8192 (16384 with argv[1] = 14) as one kernel on a 64 (128) KiB LDS tile against
strided + tile pass."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 13
n = 1 << L
ONE = 2 if L == 14 else 1
for bits, batch in ((54, 1 << (28 - L)), (49, 1 << (28 - L)), (60, 1 << (28 - L)), (28, 1 << (28 - L)), (54, 512), (54, 64)):
    q = hx.GeneratePrimes(1, bits, True, n)[0]
    ntt = hx.NTT(n, q)
    x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
    hx.fill_splitmix(x, n, batch, 1, q)
    ref = x[:2].clone()
    for t13 in (0, ONE, 0, ONE):
        hx.set_tuning("tile13", t13)
        for _ in range(5):
            ntt.ComputeForward(x, x, 1, 1)
            ntt.ComputeInverse(x, x, 1, 1)
        torch.cuda.synchronize()
        steps = 20 if batch > 1000 else 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(steps):
            ntt.ComputeForward(x, x, 1, 1)
            ntt.ComputeInverse(x, x, 1, 1)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        print("N=%d x %5d, %d-bit, %-14s %8.4f ms/step  %6.2f M NTT/s" % (
            n, batch, bits, "one kernel" if t13 else "strided + tile", ms, 2 * batch / ms / 1e3), flush=True)
    assert torch.equal(ref, x[:2])
    # bit-exactness of the one-kernel path against the two-pass path
    for fwd in (True, False):
        outs = []
        for t13 in (0, ONE):
            hx.set_tuning("tile13", t13)
            y = x[:64].clone()
            (ntt.ComputeForward if fwd else ntt.ComputeInverse)(y, y, 1, 1)
            outs.append(y)
        assert torch.equal(outs[0], outs[1]), (bits, batch, fwd)
    hx.set_tuning("tile13", 2)
