"""configs[1] (N = 4096 x 256, 50-bit prime) under rocprofv3 --kernel-trace: kernel duration
against the back-to-back per-call time (how much of a call is launch overhead)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

n, q, batch = 4096, 562949954093057, 256
ntt = hx.NTT(n, q)
x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
hx.fill_splitmix(x, n, batch, 1, q)
for fn in (ntt.ComputeForward, ntt.ComputeInverse):
    for _ in range(50):
        fn(x, x, 1, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(1000):
        fn(x, x, 1, 1)
    e1.record()
    torch.cuda.synchronize()
    print("%s: %.2f us per call back to back" % (fn.__name__, e0.elapsed_time(e1)), flush=True)
