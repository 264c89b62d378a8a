"""Where a fused_pass workgroup's time goes (needs the -DHEXL_AMD_FUSED_STATS build:
tools/build_variant.sh fstats -DHEXL_AMD_FUSED_STATS; HEXL_AMD_LIB=tools/libhexl_amd_fstats.so)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hexl_amd as hx  # noqa: E402

N, B = 65536, 4096
q = 18014398510661633
ntt = hx.NTT(N, q)
x = torch.empty((B, N), dtype=torch.int64, device="cuda")
hx.fill_splitmix(x, N, B, 1, q)
for _ in range(3):
    ntt.ComputeForward(x, x, 1, 1)
    ntt.ComputeInverse(x, x, 1, 1)
torch.cuda.synchronize()
buf = torch.zeros((4096, 8), dtype=torch.int64, device="cuda")
hx.lib.hexl_amd_debug_set_fused_stats.argtypes = [C.c_void_p]
hx.lib.hexl_amd_debug_set_fused_stats(buf.data_ptr())
for name, fn in (("forward", ntt.ComputeForward), ("inverse", ntt.ComputeInverse)):
    buf.zero_()
    torch.cuda.synchronize()
    fn(x, x, 1, 1)
    torch.cuda.synchronize()
    s = buf.cpu()
    s = s[s[:, 6] > 0].double()
    tot = s[:, 6].mean()
    print(f"{name}: {len(s)} workgroups, avg life {tot:.0f} cycles; tasks/wg phase1 {s[:,0].mean():.1f} "
          f"phase2 {s[:,1].mean():.1f}")
    print(f"   claim+barrier {100*s[:,2].mean()/tot:.1f}% (of which dependency wait "
          f"{100*s[:,3].mean()/tot:.1f}%), phase-1 bodies {100*s[:,4].mean()/tot:.1f}%, "
          f"phase-2 bodies {100*s[:,5].mean()/tot:.1f}%")
    print(f"   cycles per phase-1 task {s[:,4].sum()/max(1,s[:,0].sum()):.0f}, per phase-2 task "
          f"{s[:,5].sum()/max(1,s[:,1].sum()):.0f}, per claim {s[:,2].sum()/(s[:,0].sum()+s[:,1].sum()):.0f}")
    for xcc in range(8):
        m = s[s[:, 7] == xcc]
        if len(m):
            print(f"   xcc {xcc}: {len(m)} wgs, polys ~{m[:,1].sum()/ (32 if name=='forward' else 8):.0f}")
