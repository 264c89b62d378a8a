"""One transform shape, a few launches, for tools/pmc_one.sh: python tools/pmc_probe.py LOGN BITS BATCH WALK14 [fwd|inv|both]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx
logn, bits, batch, walk = (int(v) for v in sys.argv[1:5])
which = sys.argv[5] if len(sys.argv) > 5 else "both"
n = 1 << logn
hx.set_tuning("walk14", walk)
q = hx.GeneratePrimes(1, bits, True, n)[0]
ntt = hx.NTT(n, q)
x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
hx.fill_splitmix(x, n, batch, 1, q)
for _ in range(6):
    if which in ("fwd", "both"):
        ntt.ComputeForward(x, x, 1, 1)
    if which in ("inv", "both"):
        ntt.ComputeInverse(x, x, 1, 1)
torch.cuda.synchronize()
