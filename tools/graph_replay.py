"""Launch-bound shapes (BASELINE config 2: N=4096, 256 polynomials): the C-ABI launches are
stream-ordered and allocation-free, so a caller can capture them in a HIP graph.  Times
fwd+inv per call eagerly and replayed from a graph of 16 fwd+inv pairs."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

N, B, q = 4096, 256, 562949954093057
ntt = hx.NTT(N, q)
x = torch.empty((B, N), dtype=torch.int64, device="cuda")
hx.fill_splitmix(x, N, B, 1, q)
ref = x.clone()


def pair():
    ntt.ComputeForward(x, x, 1, 1)
    ntt.ComputeInverse(x, x, 1, 1)


for _ in range(10):
    pair()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(1600):
    pair()
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 1600
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    pair()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(16):
            pair()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    g.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / 1600
assert torch.equal(x, ref)
print(f"N={N} x {B}: fwd+inv eager {eager * 1e6:.2f} us/pair ({2 * B / eager / 1e6:.1f} M NTT/s), "
      f"HIP graph of 16 pairs {graph * 1e6:.2f} us/pair ({2 * B / graph / 1e6:.1f} M NTT/s)")
