"""Wall-clock time of the headline step (fwd + inv, N=65536, batch 4096) without
per-kernel events; for A/B runs of launch-level settings (HEXL_AMD_CHUNKS, ...)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

N, B = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
q = int(os.environ.get("NTT_Q", "18014398510661633"))
ntt = hx.NTT(N, q)
x = torch.empty((B, N), dtype=torch.int64, device="cuda")
hx.fill_splitmix(x, N, B, 1, q)
ref = x[:3].clone()
for _ in range(3):
    ntt.ComputeForward(x, x, 1, 1)
    ntt.ComputeInverse(x, x, 1, 1)
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(10):
        ntt.ComputeForward(x, x, 1, 1)
        ntt.ComputeInverse(x, x, 1, 1)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 10)
assert torch.equal(ref, x[:3])
print("chunks=%s plan=%s  %.3f ms/step  %.3f M NTT/s" % (
    os.environ.get("HEXL_AMD_CHUNKS", "-"), os.environ.get("HEXL_AMD_PLAN", "-"),
    best * 1e3, 2 * B / best / 1e6))
