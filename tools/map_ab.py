"""Per-call time of the multi-plan (prime map) transforms at N = 65536: 8 primes interleaved
(SEAL's [ciphertext][component][modulus][N] layout), `polys` polynomials per call.
HEXL_AMD_LIB selects the build.  python tools/map_ab.py [polys]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

N = 65536
polys = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for bits in (54, 59):
    primes = hx.GeneratePrimes(8, bits, True, N)
    plans = [hx.NTT(N, p) for p in primes]
    x = torch.empty((polys, N), dtype=torch.int64, device="cuda")
    for i in range(polys):
        hx.fill_splitmix(x[i], N, 1, 1 + i, primes[i % 8])
    tab = list(range(8))

    def step():
        hx.ComputeForwardMap(plans, tab, 1, x, x, 1, 1)
        hx.ComputeInverseMap(plans, tab, 1, x, x, 1, 1)
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    hx.profile_start(256)
    for _ in range(20):
        step()
    agg = {}
    for k, v in hx.profile_stop():
        agg.setdefault(k.replace("ntt_", ""), []).append(v)
    print(os.path.basename(os.environ.get("HEXL_AMD_LIB", "product")), f"{bits + 1}-bit primes, {polys} polynomials:",
          {k: round(sum(v) / len(v) * 1e3, 1) for k, v in agg.items()}, "us per launch", flush=True)
