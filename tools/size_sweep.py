"""Throughput of the forward and inverse transform over every degree 2^10 .. 2^20 and one prime per
arithmetic policy, batches of a fixed number of bytes resident in HBM (default 1 GiB).  Per cell: ms per
forward / inverse pass over the batch, launches per transform, algorithmic GB/s (16 N bytes per
transform: one read and one write of the polynomial) and its fraction of the 8 TB/s peak.  Prints a
markdown table ("prime bits" b = the first prime GeneratePrimes(1, b, true, N) returns, in (2^b, 2^(b+1)):
Small, Fp64L, Fp64, Lazy, Lazy32, Lazy16, Harvey60 and Strict arithmetic); times are HIP-event medians taken by the library's own launch profiler after a
150 ms warm-up of the same call."""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

BYTES = int(os.environ.get("SWEEP_MIB", "1024")) << 20
BITS = [int(b) for b in os.environ.get("SWEEP_BITS", "28,44,49,55,56,58,60,61").split(",")]
LOGN = range(int(os.environ.get("SWEEP_LOGN_MIN", "10")), int(os.environ.get("SWEEP_LOGN_MAX", "20")) + 1)
REPS = 15
WARM_MS = float(os.environ.get("SWEEP_WARM_MS", "150"))


def timed(fn):
    """Median over REPS of the summed kernel time of one call (ms) and the launches per call.  The
    call is first repeated for WARM_MS of wall time: a burst that starts from an idle GPU runs its
    first tens of milliseconds below the sustained rate (clock ramp), up to 15 % at these sizes."""
    import time
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < WARM_MS:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    hx.profile_start(4096)
    for _ in range(REPS):
        fn()
    torch.cuda.synchronize()
    rec = hx.profile_stop()
    per = len(rec) // REPS
    sums = [sum(v for _, v in rec[i * per:(i + 1) * per]) for i in range(REPS)]
    return statistics.median(sums), per


def main():
    print("| N | prime bits | batch | fwd ms | inv ms | launches | fwd GB/s (frac) | inv GB/s (frac) | M NTT/s (forward + inverse, each counted) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for logn in LOGN:
        n = 1 << logn
        batch = max(1, BYTES // (8 * n))
        x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
        for bits in BITS:
            q = hx.GeneratePrimes(1, bits, True, n)[0]
            ntt = hx.NTT(n, q)
            hx.fill_splitmix(x, n, batch, 1, q)
            f, lf = timed(lambda: ntt.ComputeForward(x, x, 1, 1))
            i, li = timed(lambda: ntt.ComputeInverse(x, x, 1, 1))
            gb = 16.0 * n * batch / 1e9
            print(f"| 2^{logn} | {bits} | {batch} | {f:.3f} | {i:.3f} | {lf}/{li} | "
                  f"{gb / f * 1e3:.0f} ({gb / f * 1e3 / 8000:.2f}) | {gb / i * 1e3:.0f} ({gb / i * 1e3 / 8000:.2f}) | "
                  f"{2 * batch / (f + i) * 1e-3:.2f} |", flush=True)
        del x
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
