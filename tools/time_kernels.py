"""Per-kernel times of the headline workload without correctness checks (for A/B
experiment builds selected with HEXL_AMD_LIB)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

N = int(os.environ.get("NTT_N", "65536"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
q = int(os.environ.get("NTT_Q", "18014398510661633"))
ntt = hx.NTT(N, q)
x = torch.empty((B, N), dtype=torch.int64, device="cuda")
hx.fill_splitmix(x, N, B, 1, q)
for _ in range(2):
    ntt.ComputeForward(x, x, 1, 1)
    ntt.ComputeInverse(x, x, 1, 1)
torch.cuda.synchronize()
hx.profile_start(256)
for _ in range(10):
    ntt.ComputeForward(x, x, 1, 1)
    ntt.ComputeInverse(x, x, 1, 1)
torch.cuda.synchronize()
rec = hx.profile_stop()
agg = {}
for k, v in rec:
    agg.setdefault(k, []).append(v)
REPS = 10  # (a name may be launched several times per transform: degrees above 2^17)
print(os.path.basename(os.environ.get("HEXL_AMD_LIB", "default")),
      {k.replace("ntt_", ""): round(sum(v) / len(v), 3) for k, v in agg.items()},
      "launches per step %d," % (sum(len(v) for v in agg.values()) // REPS),
      "sum %.3f ms per step" % (sum(sum(v) for v in agg.values()) / REPS))
