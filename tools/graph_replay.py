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

# ---- a composite call: KeySwitch (n = 16384, 7 decomposition moduli) is eleven launches over a
# stream-keyed scratch buffer; warmed up on the capture stream (the buffer and the plans then
# exist) it is capturable like the plain transforms.
import numpy as np  # noqa: E402

rng = np.random.default_rng(3)
n, D, C = 16384, 7, 2
K = D + 1
moduli = [int(p) for p in hx.GeneratePrimes(K, 54, True, n)]
keys = [hx.from_numpy(np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                                      for _ in range(C) for i in range(K)])) for _ in range(D)]
msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
target = hx.from_numpy(np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64) for j in range(D)]))
result = hx.from_numpy(np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                                       for _ in range(C) for i in range(D)]))


def ks(out):
    hx.KeySwitch(out, target, n, D, K, D + 1, C, moduli, keys, msf)


want = result.clone()
ks(want)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    ks(result.clone())
torch.cuda.synchronize()
eager = (time.perf_counter() - t0) / 200
outs = [result.clone() for _ in range(8)]
with torch.cuda.stream(s):
    ks(result.clone())
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=s):
        for o in outs:
            ks(o)
torch.cuda.synchronize()
for o in outs:  # (capturing records the launches; the first replay computes)
    o.copy_(result)
g2.replay()
torch.cuda.synchronize()
assert all(torch.equal(o, want) for o in outs)
t0 = time.perf_counter()
for _ in range(50):
    g2.replay()
torch.cuda.synchronize()
graph = (time.perf_counter() - t0) / 400
print(f"KeySwitch n={n}, D={D}: eager {eager * 1e6:.1f} us per call (incl. one result copy), "
      f"HIP graph of 8 calls {graph * 1e6:.1f} us per call")
