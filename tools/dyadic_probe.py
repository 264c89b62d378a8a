"""Workload for the counter passes behind the DyadicMultiply note (tools/dyadic_pmc.sh): the batched
DyadicMultiply of the bench's composites block (n = 32768 x 16 moduli, 64 pairs: 1.88 GB of algorithmic
traffic per call, 56 bytes per coefficient in seven streams) and an EltwiseMultMod over the same number
of bytes (24 bytes per element in three streams), five calls each."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

n, k, pairs = 32768, 16, 64
moduli = hx.GeneratePrimes(k, 54, True, n)
x = torch.empty(2 * n * k * pairs, dtype=torch.int64, device="cuda")
y = torch.empty_like(x)
r = torch.empty(3 * n * k * pairs, dtype=torch.int64, device="cuda")
hx.fill_splitmix(x, x.numel(), 1, 1, moduli[0])
hx.fill_splitmix(y, y.numel(), 1, 2, moduli[0])
for _ in range(5):
    hx.DyadicMultiplyBatch(r, x, y, pairs, n, moduli)
m = 56 * n * k * pairs // 24
a = torch.empty(m, dtype=torch.int64, device="cuda")
b = torch.empty_like(a)
c = torch.empty_like(a)
hx.fill_splitmix(a, m, 1, 3, moduli[0])
hx.fill_splitmix(b, m, 1, 4, moduli[0])
for _ in range(5):
    hx.EltwiseMultMod(c, a, b, m, moduli[0], 1)
torch.cuda.synchronize()
