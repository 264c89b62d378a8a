"""Workload driver for the per-kernel counter profiles of round 6 (profiles/r6_pmc_*.md).

Runs every CELL below a few times, each on device-resident data, so that ONE process under
`rocprofv3 --pmc ...` (or `--kernel-trace --stats`) sees every kernel the library ships outside the
headline configuration (whose own profile is tools/collect_profiles.sh):

  * the one-kernel transform plans: N = 4096 (BASELINE configs[1], 256 polynomials AND a 1 GiB
    batch), N = 8192 and N = 16384 at 1 GiB, one prime per arithmetic policy, both directions
  * the headline shape under the cheap policies (Small, Fp64: N = 65536 x 4096)
  * the element-wise kernels at BASELINE configs[4]'s size (N = 131072 x 1024, 61-bit) and
    MultMod at configs[1]'s
  * the composites: DyadicMultiply, KeySwitch with 256 targets per call at n = 16384

A marker launch (a tiny AddMod whose grid encodes the cell's index) precedes every cell, so the
summariser (tools/summarize_pmc_cells.py) attributes each dispatch to its cell by dispatch order and
groups counters by (cell, kernel name, grid size).  Prints a JSON manifest (the cells, their algorithmic bytes
per launch) on stdout; `--list` prints it without touching the GPU.  No oracle, no checks: the
parity tests are tests/test_gpu_parity.py.

    python tools/pmc_cells.py [--reps 3] [--warm-ms 0] [--only REGEX] > manifest.json
"""
import argparse
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

GIB = 1 << 30
# prime bits -> arithmetic policy the library picks (choose_policy, ntt_kernels.hip)
POLICY = {28: "Small", 44: "Fp64L", 49: "Fp64", 55: "Lazy", 56: "Lazy32", 58: "Lazy16", 60: "Harvey60",
          61: "Strict"}


def ntt_cells():
    cells = []
    for logn, batches in ((12, (256, GIB // (8 << 12))), (13, (GIB // (8 << 13),)), (14, (GIB // (8 << 14),))):
        for batch in batches:
            for bits in (28, 44, 49, 55, 60):
                cells.append({"kind": "ntt", "logn": logn, "bits": bits, "batch": batch,
                              "label": f"ntt N=2^{logn} {bits}-bit ({POLICY[bits]}) x {batch}",
                              "alg_bytes": 16 * (1 << logn) * batch})
    for bits in (28, 49):  # the headline shape under the cheap policies
        cells.append({"kind": "ntt", "logn": 16, "bits": bits, "batch": 4096,
                      "label": f"ntt N=2^16 {bits}-bit ({POLICY[bits]}) x 4096",
                      "alg_bytes": 16 * 65536 * 4096})
    return cells


def eltwise_cells():
    n5 = 131072 * 1024
    n2 = 4096 * 256
    return [
        {"kind": "eltwise", "op": "mult", "n": n2, "bits": 49, "label": "EltwiseMultMod configs[1] n=4096x256 50-bit",
         "alg_bytes": 24 * n2},
        {"kind": "eltwise", "op": "mult", "n": n5, "bits": 55, "label": "EltwiseMultMod n=131072x1024 55-bit",
         "alg_bytes": 24 * n5},
        {"kind": "eltwise", "op": "fma", "n": n5, "bits": 60, "label": "EltwiseFMAMod configs[4] n=131072x1024 61-bit in_mf=4",
         "alg_bytes": 24 * n5},
        {"kind": "eltwise", "op": "reduce", "n": n5, "bits": 60, "label": "EltwiseReduceMod configs[4] (q->1)",
         "alg_bytes": 16 * n5},
        {"kind": "eltwise", "op": "reduce41", "n": n5, "bits": 60, "label": "EltwiseReduceMod configs[4] (4->1)",
         "alg_bytes": 16 * n5},
        {"kind": "eltwise", "op": "reducefma", "n": n5, "bits": 60, "label": "fused ReduceMod+FMAMod configs[4]",
         "alg_bytes": 24 * n5},
        {"kind": "eltwise", "op": "add", "n": n5, "bits": 60, "label": "EltwiseAddMod n=131072x1024",
         "alg_bytes": 24 * n5},
    ]


def composite_cells():
    return [
        {"kind": "dyadic", "n": 32768, "k": 16, "reps_inner": 1, "label": "DyadicMultiply n=32768 x 16 moduli",
         "alg_bytes": 56 * 32768 * 16},
        {"kind": "keyswitch", "n": 16384, "D": 7, "bits": 54, "T": 256,
         "label": "KeySwitchBatch 256 targets n=16384 D=7 (55-bit)", "alg_bytes": None},
        {"kind": "keyswitch", "n": 8192, "D": 4, "bits": 54, "T": 256,
         "label": "KeySwitchBatch 256 targets n=8192 D=4 (55-bit)", "alg_bytes": None},
    ]


def all_cells():
    return ntt_cells() + eltwise_cells() + composite_cells()


def run(cells, reps, warm_ms=0.0):
    import time

    import numpy as np
    import torch

    import hexl_amd as hx
    rng = np.random.default_rng(1)

    def burst(fn):
        """`reps` launches of fn, behind warm_ms of the same call (the duration pass: a burst that starts from
        an idle GPU runs its first milliseconds above the sustained clock and the next ones below it while the
        power controller settles; the summariser's median then sits in the steady state)"""
        t0, calls = time.perf_counter(), 0
        while (time.perf_counter() - t0) * 1e3 < warm_ms and calls < 400:  # (the trace stays small)
            for _ in range(4):
                fn()
            calls += 4
            torch.cuda.synchronize()
        for _ in range(reps):
            fn()

    marker = torch.zeros(512 * (len(cells) + 1), dtype=torch.int64, device="cuda")
    for index, c in enumerate(cells):
        # a marker launch in front of every cell: eltwise_vec2<AddOp> over 512 * (index + 1) words, a
        # grid no measured kernel has -- the summariser attributes every later dispatch to this cell
        # (kernel names and grids alone cannot: the persistent walk's grid is the number of CUs)
        c["index"] = index
        hx.EltwiseAddMod(marker, marker, marker, 512 * (index + 1), 257)
        if c["kind"] == "ntt":
            n, batch = 1 << c["logn"], c["batch"]
            q = hx.GeneratePrimes(1, c["bits"], True, n)[0]
            x = torch.empty((batch, n), dtype=torch.int64, device="cuda")
            ntt = hx.NTT(n, q)
            hx.fill_splitmix(x, n, batch, 1, q)
            burst(lambda: ntt.ComputeForward(x, x, 1, 1))
            burst(lambda: ntt.ComputeInverse(x, x, 1, 1))
            torch.cuda.synchronize()
            c["q"] = int(q)
            del x
        elif c["kind"] == "eltwise":
            n = c["n"]
            q = hx.GeneratePrimes(1, c["bits"], True, 131072)[0]
            a = torch.empty(n, dtype=torch.int64, device="cuda")
            b = torch.empty(n, dtype=torch.int64, device="cuda")
            r = torch.empty(n, dtype=torch.int64, device="cuda")
            bound = {"mult": q, "fma": 4 * q, "reduce": 0, "reduce41": 4 * q, "reducefma": 0, "add": q}[c["op"]]
            hx.fill_splitmix(a, n, 1, 11, bound)
            hx.fill_splitmix(b, n, 1, 911, bound)
            s = 3 * q + 12345
            burst({"mult": lambda: hx.EltwiseMultMod(r, a, b, n, q, 1),
                   "fma": lambda: hx.EltwiseFMAMod(r, a, s, b, n, q, 4),
                   "reduce": lambda: hx.EltwiseReduceMod(r, a, n, q, q, 1),
                   "reduce41": lambda: hx.EltwiseReduceMod(r, a, n, q, 4, 1),
                   "reducefma": lambda: hx.EltwiseReduceFMAMod(r, a, s % q, b, n, q, q),
                   "add": lambda: hx.EltwiseAddMod(r, a, b, n, q)}[c["op"]])
            torch.cuda.synchronize()
            c["q"] = int(q)
            del a, b, r
        elif c["kind"] == "dyadic":
            n, k = c["n"], c["k"]
            moduli = [int(q) for q in hx.GeneratePrimes(k, 54, True, n)]
            x = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli] * 2)
            y = np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli] * 2)
            dx, dy = hx.from_numpy(x), hx.from_numpy(y)
            out = hx.from_numpy(np.zeros(3 * n * k, dtype=np.uint64))
            burst(lambda: hx.DyadicMultiply(out, dx, dy, n, moduli))
            torch.cuda.synchronize()
        elif c["kind"] == "keyswitch":
            n, D, T = c["n"], c["D"], c["T"]
            K, C = D + 1, 2
            moduli = [int(q) for q in hx.GeneratePrimes(K, c["bits"], True, n)]
            target = np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64) for j in range(D)])
            keys = [np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                                    for _ in range(C) for i in range(K)]) for _ in range(D)]
            msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
            result = np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                                     for _ in range(C) for i in range(D)])
            d_keys = [hx.from_numpy(kk) for kk in keys]
            d_tt = hx.from_numpy(np.tile(target, T))
            d_rr = hx.from_numpy(np.tile(result, T))
            burst(lambda: hx.KeySwitchBatch(d_rr, d_tt, T, n, D, K, D + 1, C, moduli, d_keys, msf))
            torch.cuda.synchronize()
            # algorithmic bytes of the whole call: targets in (D polys) + result in and out
            # (C * D polys each) per target; keys D * C * K polys once (shared by the targets)
            c["alg_bytes"] = 8 * n * (T * (D + 2 * C * D) + D * C * K)
            del d_tt, d_rr
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--warm-ms", type=float, default=0.0)
    ap.add_argument("--only", default=None)
    ap.add_argument("--list", action="store_true")
    a = ap.parse_args()
    cells = all_cells()
    if a.only:
        cells = [c for c in cells if re.search(a.only, c["label"])]
    if not a.list:
        run(cells, a.reps, a.warm_ms)
    json.dump({"reps": a.reps, "warm_ms": a.warm_ms, "cells": cells}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
