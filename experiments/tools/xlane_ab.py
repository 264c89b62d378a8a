"""A/B of the in-wave round hand-over: through LDS (product) against the cross-lane exchange
(-DHEXL_AMD_XLANE=1: v_permlane32_swap / v_permlane16_swap / DPP row_ror:8), in the three
regimes VERDICT r2 item 4 names: BASELINE configs[1] (N = 4096 x 256, one tile per CU: a serial
latency chain), the Fp64 tile pass at the headline shape (HBM-bound, VALU slack) and the Lazy
tile pass (VALU-bound).  Run once per library: HEXL_AMD_LIB=tools/libhexl_amd_xlane.so."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

name = os.path.basename(os.environ.get("HEXL_AMD_LIB", "product (LDS hand-over)"))


def per_call_us(fn, iters):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def kernels(ntt, x, reps):
    for _ in range(6):
        ntt.ComputeForward(x, x, 1, 1)
        ntt.ComputeInverse(x, x, 1, 1)
    torch.cuda.synchronize()
    hx.profile_start(8 * reps + 8)
    for _ in range(reps):
        ntt.ComputeForward(x, x, 1, 1)
        ntt.ComputeInverse(x, x, 1, 1)
    torch.cuda.synchronize()
    agg = {}
    for k, v in hx.profile_stop():
        agg.setdefault(k.replace("ntt_", ""), []).append(v)
    return {k: round(sorted(v)[len(v) // 2], 4) for k, v in agg.items()}


# (i) configs[1]: N = 4096, 50-bit prime (Fp64 policy), 256 polynomials; and a 55-bit prime (Lazy)
for q in (562949954093057, hx.GeneratePrimes(1, 54, True, 4096)[0]):
    ntt = hx.NTT(4096, q)
    x = torch.empty((256, 4096), dtype=torch.int64, device="cuda")
    y = torch.empty_like(x)
    hx.fill_splitmix(x, 4096, 256, 1, q)
    f = per_call_us(lambda: ntt.ComputeForward(y, x, 1, 1), 400)
    i = per_call_us(lambda: ntt.ComputeInverse(y, x, 1, 1), 400)
    print(f"{name}: N=4096 x 256, q={q}: forward {f:.2f} us, inverse {i:.2f} us per call "
          f"(median kernel ms {kernels(ntt, x.clone(), 50)})")
# (ii) / (iii) headline shape, Fp64 (50-bit prime) and Lazy (55-bit prime)
x = torch.empty((4096, 65536), dtype=torch.int64, device="cuda")
for label, q in (("Fp64, 50-bit", hx.GeneratePrimes(1, 49, True, 65536)[0]),
                 ("Lazy, 55-bit", 18014398510661633)):
    ntt = hx.NTT(65536, q)
    hx.fill_splitmix(x, 65536, 4096, 1, q)
    for _ in range(10):
        ntt.ComputeForward(x, x, 1, 1)
        ntt.ComputeInverse(x, x, 1, 1)
    print(f"{name}: N=65536 x 4096 ({label}): median kernel ms {kernels(ntt, x, 10)}")
