"""KeySwitchBatch(256 targets, n = 16384, D = 7) for `rocprofv3 --kernel-trace --stats`: which of the
launches the per-target time goes to.  (Under tests/ because it shares the oracle-checked setup of
bench_composites.py; not collected by pytest.)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

rng = np.random.default_rng(1)
n, D, T = (int(a) for a in (sys.argv[1:4] + ["16384", "7", "256"][len(sys.argv) - 1:]))
K, C = D + 1, 2
moduli = [int(q) for q in hx.GeneratePrimes(K, 54, True, n)]
keys = [hx.from_numpy(np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                                      for _ in range(C) for i in range(K)])) for _ in range(D)]
msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
tt = hx.from_numpy(np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64)
                                   for _ in range(T) for j in range(D)]))
rr = hx.from_numpy(np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                                   for _ in range(T) for _ in range(C) for i in range(D)]))
for _ in range(12):
    hx.KeySwitchBatch(rr, tt, T, n, D, K, D + 1, C, moduli, keys, msf)
torch.cuda.synchronize()
