"""Developer diagnostic: where does a block_pass workgroup spend its life?

Loads tools/libhexl_amd_phaseprof.so (the product sources built with
`tools/build_variant.sh phaseprof -DHEXL_AMD_PHASE_PROFILE`), runs the forward NTT at
the headline shape and prints, per phase, the mean / p50 / p90 number of shader
cycles a wave spends between consecutive stamps.
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.environ.get("PHASE_LIB") or os.path.join(HERE, "libhexl_amd_phaseprof.so"))
vp, u64 = C.c_void_p, C.c_uint64
lib.hexl_amd_ntt_create.argtypes = [C.POINTER(vp), u64, u64, u64, C.c_int]
lib.hexl_amd_ntt_forward.argtypes = [vp, vp, vp, u64, u64, u64, vp]
lib.hexl_amd_ntt_inverse.argtypes = [vp, vp, vp, u64, u64, u64, vp]
lib.hexl_amd_debug_set_phase_buf.argtypes = [vp]
lib.hexl_amd_last_error.restype = C.c_char_p

N, BATCH = int(os.environ.get("PHASE_N", "65536")), int(sys.argv[1]) if len(sys.argv) > 1 else 4096
Q = int(os.environ.get("PHASE_Q", "18014398510661633"))
INVERSE = os.environ.get("PHASE_DIR", "fwd") == "inv"  # stamps of the inverse tile pass instead
plan = vp()
assert lib.hexl_amd_ntt_create(C.byref(plan), N, Q, 0, 0) == 0, lib.hexl_amd_last_error()
data = torch.randint(0, Q, (BATCH, N), dtype=torch.int64, device="cuda")
TILE = int(os.environ.get("PHASE_TILE", "2048"))  # elements per workgroup of the tile pass
blocks = BATCH * N // TILE
stamps = torch.zeros((blocks, 8, 16), dtype=torch.int64, device="cuda")
for it in range(3):
    stamps.zero_()
    torch.cuda.synchronize()
    lib.hexl_amd_debug_set_phase_buf(vp(stamps.data_ptr()))
    if INVERSE:
        rc = lib.hexl_amd_ntt_inverse(plan, vp(data.data_ptr()), vp(data.data_ptr()), BATCH, 1, 1, None)
    else:
        rc = lib.hexl_amd_ntt_forward(plan, vp(data.data_ptr()), vp(data.data_ptr()), BATCH, 4, 4, None)
    assert rc == 0, lib.hexl_amd_last_error()
    torch.cuda.synchronize()
    lib.hexl_amd_debug_set_phase_buf(None)
s = stamps.cpu().numpy().astype(np.int64)
s = s[:, :TILE // 512, :]  # waves a workgroup really has
names = {1: "global loads landed", 2: "round 0 compute", 3: "LDS store + block barrier",
         4: "round 1 (load, compute, store, wave sync)", 5: "round 2", 6: "round 3",
         8: "copy-out: LDS read, finish, issue stores", 9: "stores acknowledged"}
order = [0, 1, 2, 3, 4, 5, 6, 8, 9]
if INVERSE:
    names = {1: "global loads landed", 2: "LDS copy-in + wave sync", 3: "deepest round (load, compute, store, sync)",
             4: "next round", 5: "next round (+ block barrier)", 6: "next round", 7: "round 0: LDS load, compute",
             8: "finish, issue stores", 9: "stores acknowledged"}
    order = [i for i in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9) if s[:, :, i].max() > 0]
order = [i for i in order if i == 0 or s[:, :, i].max() > 0]
print(f"{'inverse' if INVERSE else 'forward'} tile pass, N {N}, batch {BATCH}: per-wave cycles between stamps")
tot = s[:, :, 9] - s[:, :, 0]
for a, b in zip(order[:-1], order[1:]):
    d = (s[:, :, b] - s[:, :, a]).reshape(-1)
    print(f"  {names[b]:48s} mean {d.mean():9.0f}  p50 {np.percentile(d, 50):9.0f}  "
          f"p90 {np.percentile(d, 90):9.0f}  ({100.0 * d.mean() / tot.mean():4.1f}%)")
print(f"  wave lifetime mean {tot.mean():.0f} cycles; kernel span "
      f"{(s[:, :, 9].max() - s[:, :, 0][s[:, :, 0] > 0].min())} cycles")
