"""Developer check of the persistent 14-stage tile walk (set_tuning("walk14", 1)) against the
one-workgroup-per-tile kernels ("walk14", 0): bit-identical outputs, every arithmetic policy, both
directions, a batch that is not a multiple of the grid, and the multi-plan (RNS) form."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx

n = 16384
ok = True
for bits in (28, 44, 49, 55, 56, 58, 60, 61):
    q = hx.GeneratePrimes(1, bits, True, n)[0]
    ntt = hx.NTT(n, q)
    for batch in (300, 777, 1024):
        x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
        hx.fill_splitmix(x, n, batch, 3, q)
        res = {}
        for walk in (0, 1):
            hx.set_tuning("walk14", walk)
            f = torch.empty_like(x)
            ntt.ComputeForward(f, x, 1, 1)
            i = torch.empty_like(x)
            ntt.ComputeInverse(i, f, 1, 1)
            f4 = x.clone()
            ntt.ComputeForward(f4, f4, 4, 4)  # in place, lazy output
            i2 = f.clone()
            ntt.ComputeInverse(i2, i2, 2, 2)
            res[walk] = (f, i, f4 % q, i2 % q)
        same = all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
        rt = torch.equal(res[1][1], x)
        print(bits, batch, "same" if same else "DIFFERENT", "round-trip ok" if rt else "ROUND TRIP BROKEN", flush=True)
        ok = ok and same and rt
# multi-plan: RNS limbs, several moduli in one launch
moduli = hx.GeneratePrimes(4, 54, True, n)
plans = [hx.NTT(n, q) for q in moduli]
per = 120
x = torch.empty((len(moduli) * per, n), dtype=torch.int64, device="cuda")
for k, q in enumerate(moduli):
    hx.fill_splitmix(x[k * per:(k + 1) * per], n, per, 11 + k, q)
res = {}
for walk in (0, 1):
    hx.set_tuning("walk14", walk)
    f = torch.empty_like(x)
    hx.ComputeForwardRNS(plans, f, x, 1, 1)
    i = torch.empty_like(x)
    hx.ComputeInverseRNS(plans, i, f, 1, 1)
    res[walk] = (f, i)
same = all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
print("rns", "same" if same else "DIFFERENT", "round-trip ok" if torch.equal(res[1][1], x) else "ROUND TRIP BROKEN")
ok = ok and same and torch.equal(res[1][1], x)
hx.set_tuning("walk14", 1)
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 1)
