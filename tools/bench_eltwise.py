"""Achieved HBM bandwidth of the element-wise kernels and the secondary NTT configs
(BASELINE.json configs[1], [4]); algorithmic bytes per element per BASELINE.md section 4."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402


def timed(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


out = {}
# config 5 shape: N=131072 x 1024 = 2^27 elements, 61-bit prime (q < 2^61 for FMAMod)
q = hx.GeneratePrimes(1, 60, True, 131072)[0]
n = 131072 * 1024
a = torch.empty(n, dtype=torch.int64, device="cuda")
b = torch.empty(n, dtype=torch.int64, device="cuda")
r = torch.empty(n, dtype=torch.int64, device="cuda")
hx.fill_splitmix(a, 131072, 1024, 1, q)
hx.fill_splitmix(b, 131072, 1024, 5001, q)
ops = {
    "EltwiseAddMod (24 B/elt)": (lambda: hx.EltwiseAddMod(r, a, b, n, q), 24),
    "EltwiseAddMod scalar (16 B/elt)": (lambda: hx.EltwiseAddMod(r, a, 12345, n, q), 16),
    "EltwiseSubMod (24 B/elt)": (lambda: hx.EltwiseSubMod(r, a, b, n, q), 24),
    "EltwiseMultMod in_mf=1 (24 B/elt)": (lambda: hx.EltwiseMultMod(r, a, b, n, q, 1), 24),
    "EltwiseMultMod in_mf=4 (24 B/elt)": (lambda: hx.EltwiseMultMod(r, a, b, n, q, 4), 24),
    "EltwiseFMAMod in_mf=4 +addend (24 B/elt)": (lambda: hx.EltwiseFMAMod(r, a, 777, b, n, q, 4), 24),
    "EltwiseFMAMod no addend (16 B/elt)": (lambda: hx.EltwiseFMAMod(r, a, 777, None, n, q, 1), 16),
    "EltwiseReduceMod q->1 (16 B/elt)": (lambda: hx.EltwiseReduceMod(r, a, n, q, q, 1), 16),
    "EltwiseReduceMod 4->1 (16 B/elt)": (lambda: hx.EltwiseReduceMod(r, a, n, q, 4, 1), 16),
    "fused ReduceMod+FMAMod (24 B/elt)": (lambda: hx.EltwiseReduceFMAMod(r, a, 777, b, n, q, q), 24),
}
for name, (fn, bpe) in ops.items():
    t = timed(fn)
    out[name] = {"ms": t * 1e3, "GBps": n * bpe / t / 1e9, "Gelt_per_s": n / t / 1e9}
    print(f"{name:45s} {t*1e3:8.3f} ms  {n*bpe/t/1e9:8.1f} GB/s")
del a, b, r
# config 2: N=4096, 50-bit, batch 256: fwd / inv / multmod
N, B = 4096, 256
q = hx.GeneratePrimes(1, 49, True, N)[0]
x = torch.empty((B, N), dtype=torch.int64, device="cuda")
hx.fill_splitmix(x, N, B, 1, q)
y = x.clone()
ntt = hx.NTT(N, q)
for name, fn in (("config2 fwd NTT N=4096 x256", lambda: ntt.ComputeForward(y, x, 1, 1)),
                 ("config2 inv NTT N=4096 x256", lambda: ntt.ComputeInverse(y, x, 1, 1)),
                 ("config2 MultMod 4096x256", lambda: hx.EltwiseMultMod(y, x, x, N * B, q, 1))):
    t = timed(fn, 50)
    out[name] = {"us": t * 1e6, "NTT_per_s": B / t}
    print(f"{name:45s} {t*1e6:8.2f} us  ({B/t/1e6:.2f} M polys/s)")
# config 5 NTT: N=131072, 61-bit, batch 1024 (Strict arithmetic policy)
N, B = 131072, 1024
q = hx.GeneratePrimes(1, 60, True, N)[0]
x = torch.empty((B, N), dtype=torch.int64, device="cuda")
hx.fill_splitmix(x, N, B, 1, q)
ntt = hx.NTT(N, q)
for name, fn in (("N=131072 61-bit fwd x1024", lambda: ntt.ComputeForward(x, x, 1, 1)),
                 ("N=131072 61-bit inv x1024", lambda: ntt.ComputeInverse(x, x, 1, 1))):
    t = timed(fn, 5)
    out[name] = {"ms": t * 1e3, "NTT_per_s": B / t, "GBps_algorithmic": B * 16 * N / t / 1e9}
    print(f"{name:45s} {t*1e3:8.3f} ms  ({B/t/1e3:.1f} k NTT/s, {B*16*N/t/1e9:.0f} GB/s algorithmic)")
print(json.dumps(out))
