"""Per-kernel time of the forward + inverse transform at the headline shape (N = 65536 x 4096) for one prime per
arithmetic policy, under the library build HEXL_AMD_LIB selects: median HIP-event duration per kernel family over 15
repetitions after a warm-up.  python tools/ab_headline.py [bits,...]"""
import os, sys, statistics, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx
bits = [int(b) for b in (sys.argv[1] if len(sys.argv) > 1 else "28,49,55").split(",")]
n, batch = 65536, 4096
x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
for b in bits:
    q = hx.GeneratePrimes(1, b, True, n)[0]
    ntt = hx.NTT(n, q)
    hx.fill_splitmix(x, n, batch, 1, q)
    def step():
        ntt.ComputeForward(x, x, 1, 1)
        ntt.ComputeInverse(x, x, 1, 1)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        step()
    torch.cuda.synchronize()
    hx.profile_start(256)
    for _ in range(15):
        step()
    torch.cuda.synchronize()
    agg = {}
    for k, v in hx.profile_stop():
        agg.setdefault(k.replace("ntt_", ""), []).append(v)
    print(b, " ".join(f"{k}={statistics.median(v):.3f}" for k, v in agg.items()),
          f"step={sum(statistics.median(v) for v in agg.values()):.3f}", flush=True)
