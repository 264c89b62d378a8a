"""One ciphertext per KeySwitch call (n = 16384, D = 7, C = 2), launch by launch and replayed from the graph:
the workload for a `rocprofv3 --kernel-trace` of the dependent chain (tools/ks_chain_from_trace.py reads the
trace: per kernel duration and the gap to its predecessor)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

rng = np.random.default_rng(1)
n, D, C = int(os.environ.get("KS_N", 16384)), int(os.environ.get("KS_D", 7)), 2
K = D + 1
moduli = hx.GeneratePrimes(K, 54, True, n)
keys = [hx.from_numpy(np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                                      for _ in range(C) for i in range(K)])) for _ in range(D)]
msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
target = hx.from_numpy(np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64) for j in range(D)]))
result = hx.from_numpy(np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64) for _ in range(C) for i in range(D)]))
for graph in (0, 1):
    hx.set_tuning("ks_graph", graph)
    for _ in range(5):
        hx.KeySwitch(result, target, n, D, K, D + 1, C, moduli, keys, msf)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hx.KeySwitch(result, target, n, D, K, D + 1, C, moduli, keys, msf)
    e1.record()
    torch.cuda.synchronize()
    print(f"ks_graph={graph}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call", flush=True)
