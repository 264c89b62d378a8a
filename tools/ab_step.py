"""Same-box A/B of two builds of the library on the headline step (HEXL_AMD_LIB selects the
build): alternates `rounds` times between the libraries, each time a fresh process that runs
the fwd+inv step over 4096 polynomials of N = 65536 for ~1.5 s and prints the median HIP-event
step and the per-kernel averages.  python tools/ab_step.py libA.so libB.so [rounds] [q]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json, time
import torch
sys.path.insert(0, %r)
import hexl_amd as hx
N, B = 65536, 4096
q = int(os.environ.get("NTT_Q", "18014398510661633"))
ntt = hx.NTT(N, q)
x = torch.empty((B, N), dtype=torch.int64, device="cuda")
hx.fill_splitmix(x, N, B, 1, q)
ref = x[:2].clone()
def step():
    ntt.ComputeForward(x, x, 1, 1)
    ntt.ComputeInverse(x, x, 1, 1)
for _ in range(40): step()
torch.cuda.synchronize()
ms = []
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.5:
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(33)]
    ev[0].record()
    for i in range(32):
        step(); ev[i + 1].record()
    torch.cuda.synchronize()
    ms += [ev[i].elapsed_time(ev[i + 1]) for i in range(32)]
assert torch.equal(ref, x[:2])
hx.profile_start(256)
for _ in range(10): step()
torch.cuda.synchronize()
agg = {}
for k, v in hx.profile_stop(): agg.setdefault(k.replace("ntt_", ""), []).append(v)
ms.sort()
print(json.dumps({"median_ms": ms[len(ms) // 2], "kern": {k: round(sum(v) / len(v), 4) for k, v in agg.items()}}))
''' % ROOT

libs = sys.argv[1:3]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, HEXL_AMD_LIB=os.path.abspath(l))
        if len(sys.argv) > 4:
            env["NTT_Q"] = sys.argv[4]
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = [x for x in out.stdout.splitlines() if x.startswith("{")]
        if not line:
            print(l, "FAILED", out.stderr[-400:])
            continue
        d = json.loads(line[-1])
        res[l].append(d)
        print(os.path.basename(l), d, flush=True)
for l in libs:
    if res[l]:
        m = sorted(d["median_ms"] for d in res[l])
        print(os.path.basename(l), "median of medians %.4f ms" % m[len(m) // 2])
