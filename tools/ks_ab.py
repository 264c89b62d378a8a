"""KeySwitchBatch per-target time (n = 16384, D = 7, C = 2, 256 targets) and the kernels of one
call, for A/B builds (HEXL_AMD_LIB)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

rng = np.random.default_rng(1)
n, D, C, T = 16384, 7, 2, 256
K = D + 1
moduli = hx.GeneratePrimes(K, 54, True, n)
keys = [hx.from_numpy(np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                                      for _ in range(C) for i in range(K)])) for _ in range(D)]
msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
target = np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64) for j in range(D)])
result = np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64) for _ in range(C) for i in range(D)])
d_tt, d_rr = hx.from_numpy(np.tile(target, T)), hx.from_numpy(np.tile(result, T))


def call():
    hx.KeySwitchBatch(d_rr, d_tt, T, n, D, K, D + 1, C, moduli, keys, msf)


if os.environ.get("KS_FUSE") is not None:
    hx.set_tuning("ks_fuse", int(os.environ["KS_FUSE"]))
if os.environ.get("KS_ONESTEP") is not None:
    hx.set_tuning("ks_mac_onestep", int(os.environ["KS_ONESTEP"]))
for _ in range(3):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    call()
e1.record()
torch.cuda.synchronize()
print(os.path.basename(os.environ.get("HEXL_AMD_LIB", "product")), "ks_fuse=" + os.environ.get("KS_FUSE", "default"), "onestep=" + os.environ.get("KS_ONESTEP", "default"),
      f"{e0.elapsed_time(e1) / 10 * 1e3 / T:.2f} us per target ({e0.elapsed_time(e1) / 10:.3f} ms per call)",
      "checksum", int(d_rr.view(-1)[::4099].sum().item()) & 0xffffffff, flush=True)
