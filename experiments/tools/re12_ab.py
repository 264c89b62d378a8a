"""A/B of the 12-stage tile pass's geometry: 512 threads x 8 elements (product) against
1024 threads x 4 elements (tools/build_variant.sh re12_2 -DHEXL_AMD_RE12=2), one library
per process: `python tools/re12_ab.py [path/to/lib]`.  Prints configs[1]'s per-call
latency, its throughput shape and N = 2^17, and a checksum to compare between the two."""
import os
import sys

if len(sys.argv) > 1:
    os.environ["HEXL_AMD_LIB"] = os.path.abspath(sys.argv[1])
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402


def run(n, q, batch, steps):
    ntt = hx.NTT(n, q)
    x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
    hx.fill_splitmix(x, n, batch, 1, q)
    ref = x.clone()
    y = x.clone()
    ntt.ComputeForward(y, y, 1, 1)
    digest = int(y.sum().item()) & 0xFFFFFFFFFFFF
    ntt.ComputeInverse(y, y, 1, 1)
    assert torch.equal(y, ref)
    res = []
    for fn in (ntt.ComputeForward, ntt.ComputeInverse):
        for _ in range(20):
            fn(x, x, 1, 1)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                fn(x, x, 1, 1)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / steps)
        res.append(best * 1e3)
    print("N=%6d x %5d q=%d: fwd %8.2f us  inv %8.2f us   digest %012x" % (
        n, batch, q, res[0], res[1], digest), flush=True)


print("library:", os.environ.get("HEXL_AMD_LIB", "product"))
run(4096, 562949954093057, 256, 400)     # configs[1]
run(4096, 562949954093057, 65536, 10)    # same transform, throughput shape
run(4096, hx.GeneratePrimes(1, 54, True, 4096)[0], 256, 400)
run(4096, hx.GeneratePrimes(1, 54, True, 4096)[0], 65536, 10)
run(131072, 1152921504616808449, 1024, 10)  # configs[4]'s transform (5 + 12 stages)
