"""RNS transform (multi-plan launch) at N = 16384: 8 moduli x 512 polynomials, forward / inverse ms per call under
set_tuning("walk14", WALK)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx
n, per = 16384, 512
for bits in (54, 59, 44):
    moduli = hx.GeneratePrimes(8, bits, True, n)
    plans = [hx.NTT(n, q) for q in moduli]
    x = torch.empty((8 * per, n), dtype=torch.int64, device="cuda")
    for k, q in enumerate(moduli):
        hx.fill_splitmix(x[k * per:(k + 1) * per], n, per, 11 + k, q)
    for walk in (0, 1, 0, 1):
        hx.set_tuning("walk14", walk)
        res = []
        for fn in (hx.ComputeForwardRNS, hx.ComputeInverseRNS):
            for _ in range(20):
                fn(plans, x, x, 1, 1)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(21)]
            ev[0].record()
            for i in range(20):
                fn(plans, x, x, 1, 1)
                ev[i + 1].record()
            torch.cuda.synchronize()
            res.append(statistics.median(ev[i].elapsed_time(ev[i + 1]) for i in range(20)))
        print(f"{bits + 1}-bit walk14={walk}: fwd {res[0]:.3f} ms  inv {res[1]:.3f} ms", flush=True)
hx.set_tuning("walk14", 1)
