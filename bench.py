"""bench.py -- headline benchmark of the MI355X-native HEXL hot path.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: an
in-place forward NTT (input_mod_factor = output_mod_factor = 1) followed by an
in-place inverse NTT over 4096 polynomials of degree N = 65536 with a 55-bit
prime (BASELINE.json configs[2]; at --gpus G > 1 rank g transforms the 4096
polynomials of RNS prime g, configs[3]: embarrassingly parallel, no data-path
collective, weak scaling).  Inputs are generated on the device (splitmix64) and
are resident in HBM before the timed region starts.

Prints ONE JSON line on rank 0 (see the contract in the task statement):
value = Fwd+Inv NTTs per second over the whole job; `roofline` = algorithmic
bytes (16*N per transform) of the dominant kernel per launch /
its average duration, measured with HIP events on the launch stream inside the
timed region; `cpu_baseline` = the oracle's scalar Harvey NTT (a port of the
reference's native path, oracle/hexl_oracle.c) on this box's host cores.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N = 65536
BATCH = 4096
# GeneratePrimes(8, 54, true, 65536): the 8 RNS primes of BASELINE configs[3]
# (tests/golden/hexl_kat.json generate_primes_survey_probe)
PRIMES = [18014398510661633, 18014398512365569, 18014398514200577, 18014398514987009,
          18014398515511297, 18014398516559873, 18014398521016321, 18014398524424193]
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
TRAFFIC_PROFILE = "r2_hbm_traffic.json"
# what keeps each kernel family below the HBM roofline (profiles/r2_pmc_summary.md)
KERNEL_LIMITER = {"ntt_fwd_strided_pass": "hbm", "ntt_inv_strided_pass": "hbm",
                  "ntt_fwd_tile_pass_bottom": "valu-issue", "ntt_inv_tile_pass_bottom": "valu-issue",
                  "ntt_fwd_fused_pass": "valu-issue + scheduler", "ntt_inv_fused_pass": "valu-issue + scheduler"}
PREWARM = 20            # untimed passes before the --warmup ones (see main)


def cpu_baseline(seconds_single=4.0, seconds_all=8.0):
    """Oracle (port of the reference's native radix-2 path) on the host cores.

    Bounded sample: single-thread for ~4 s, then one worker thread per usable logical
    CPU for ~8 s, each worker looping fwd+inv over its own polynomial (HEXL itself is
    single-threaded: independent polynomials per thread is its faithful multi-core use).
    The 8-lane AVX-512 variant of the same algorithm (oracle/hexl_oracle_avx512.c:
    all stages vectorised, depth first in L1-sized blocks like the reference's
    production path; bit-identical outputs) is used when the host has AVX-512 F/DQ;
    the scalar figure is reported beside it.
    """
    import ctypes as C

    import numpy as np

    from oracle import hexl_oracle as ho

    q = PRIMES[0]
    plan = ho.lib.ho_ntt_create(N, q, 0)
    logical = os.cpu_count() or 1
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or logical
    except Exception:  # noqa: BLE001 -- psutil is optional
        physical = logical
    try:
        logical = len(os.sched_getaffinity(0))  # what this process may actually use
    except (AttributeError, OSError):
        pass
    cores = min(logical, 256)  # one worker thread per usable logical CPU
    simd = bool(ho.lib.ho_has_avx512())
    scalar_fns = (ho.lib.ho_ntt_forward_batch, ho.lib.ho_ntt_inverse_batch)
    simd_fns = (ho.lib.ho_ntt_forward_batch_avx512, ho.lib.ho_ntt_inverse_batch_avx512)

    def worker(fns, seed, deadline, counts, idx):
        fwd, inv = fns
        buf = ho.fill_splitmix(N, seed, q)
        p = buf.ctypes.data_as(C.POINTER(C.c_uint64))
        done = 0
        while time.perf_counter() < deadline:
            for _ in range(4):
                fwd(plan, p, p, 1, 1, 1)
                inv(plan, p, p, 1, 1, 1)
            done += 8
        counts[idx] = done

    def run(fns, nthreads, seconds):
        counts = [0] * nthreads
        t0 = time.perf_counter()
        deadline = t0 + seconds
        ts = [threading.Thread(target=worker, args=(fns, 1 + i, deadline, counts, i))
              for i in range(nthreads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return sum(counts) / (time.perf_counter() - t0)

    best = simd_fns if simd else scalar_fns
    single = run(best, 1, seconds_single)
    allc = run(best, cores, seconds_all)
    out = {
        "value": allc, "unit": "NTT/s", "cores": cores, "kind": "port",
        "threads": cores, "logical_cpus": logical, "physical_cores": physical,
        "isa": "avx512 (8 lanes)" if simd else "scalar",
        "single_thread_value": single,
        "sample": (f"oracle Harvey radix-2 fwd+inv NTT ({'AVX-512 variant' if simd else 'scalar'}), "
                   f"N={N}, q={q} (55-bit), 1 poly per thread in place; 1 thread x "
                   f"{seconds_single:.0f} s then {cores} threads x {seconds_all:.0f} s"),
    }
    if simd:
        out["scalar_single_thread_value"] = run(scalar_fns, 1, 2.0)
    ho.lib.ho_ntt_destroy(plan)
    return out


def event_timed(torch, fn, iters):
    """average seconds per call of fn, HIP events on the current (= launch) stream"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def rate(bytes_algorithmic, seconds, **extra):
    g = bytes_algorithmic / seconds / 1e9
    return dict({"us": seconds * 1e6, "GBps_algorithmic": g, "frac_of_hbm_peak": g / HBM_PEAK_GBPS},
                **extra)


def timed_eltwise(hx, torch, name, n, batch, q):
    a = torch.empty(batch * n, dtype=torch.int64, device="cuda")
    b = torch.empty_like(a)
    r = torch.empty_like(a)
    hx.fill_splitmix(a, n, batch, 3, q)
    hx.fill_splitmix(b, n, batch, 7, q)
    t = event_timed(torch, lambda: hx.EltwiseMultMod(r, a, b, batch * n, q, 1), 5)
    return dict(rate(24.0 * batch * n, t), op=name, elements=batch * n, ms=t * 1e3)


def secondary_configs(hx, torch):
    """BASELINE.json configs[1] and configs[4] (not the timed metric): per-call time with the
    calls issued back to back on one stream (for the 8-us config-2 kernels that is the
    launch-bound rate a caller sees)."""
    out = {}
    n, b, q = 4096, 256, 562949954093057  # configs[1]: N=4096, 50-bit prime, batch 256
    x = torch.empty((b, n), dtype=torch.int64, device="cuda")
    y = torch.empty_like(x)
    hx.fill_splitmix(x, n, b, 1, q)
    ntt = hx.NTT(n, q)
    out["config2"] = {
        "shape": f"N={n}, q={q} (50-bit), batch={b}",
        "fwd": rate(16.0 * n * b, event_timed(torch, lambda: ntt.ComputeForward(y, x, 1, 1), 200)),
        "inv": rate(16.0 * n * b, event_timed(torch, lambda: ntt.ComputeInverse(y, x, 1, 1), 200)),
        "multmod": rate(24.0 * n * b,
                        event_timed(torch, lambda: hx.EltwiseMultMod(y, x, x, n * b, q, 1), 200)),
    }
    n, b, q = 131072, 1024, 1152921504616808449  # configs[4]: N=131072, 61-bit prime, batch 1024
    a = torch.empty(b * n, dtype=torch.int64, device="cuda")
    c = torch.empty_like(a)
    r = torch.empty_like(a)
    hx.fill_splitmix(a, n, b, 21, 4 * q)
    hx.fill_splitmix(c, n, b, 1021, 4 * q)
    s = 3 * q + 12345
    e = float(n) * b
    ntt = hx.NTT(n, q)
    cfg5 = {"shape": f"N={n}, q={q} (61-bit), batch={b}"}
    cfg5["fma"] = rate(24 * e, event_timed(torch, lambda: hx.EltwiseFMAMod(r, a, s, c, b * n, q, 4), 5),
                       op="EltwiseFMAMod(input_mod_factor=4, arg3 != null)")
    cfg5["reduce"] = rate(16 * e, event_timed(torch, lambda: hx.EltwiseReduceMod(r, a, b * n, q, 4, 1), 5),
                          op="EltwiseReduceMod(4 -> 1)")
    cfg5["reduce_q"] = rate(16 * e, event_timed(torch, lambda: hx.EltwiseReduceMod(r, a, b * n, q, q, 1), 5),
                            op="EltwiseReduceMod(q -> 1)")
    cfg5["fused"] = rate(24 * e, event_timed(torch, lambda: hx.EltwiseReduceFMAMod(r, a, s % q, c, b * n, q, q), 5),
                         op="fused ReduceMod(q -> 1) + FMAMod, one kernel")
    hx.EltwiseReduceMod(a, a, b * n, q, 4, 1)
    cfg5["ntt_fwd"] = rate(16 * e, event_timed(torch, lambda: ntt.ComputeForward(r, a, 1, 1), 5))
    cfg5["ntt_inv"] = rate(16 * e, event_timed(torch, lambda: ntt.ComputeInverse(r, a, 1, 1), 5))
    out["config5"] = cfg5
    del a, c, r
    # the headline shape under the other arithmetic policies (moduli of other sizes)
    n, b = N, 4096
    x = torch.empty((b, n), dtype=torch.int64, device="cuda")
    other = {}
    for label, bits in (("30-bit prime (Small policy)", 29), ("50-bit prime (Fp64 policy)", 49),
                        ("60-bit prime (Harvey60 policy)", 59), ("62-bit prime (Strict policy)", 61)):
        q = hx.GeneratePrimes(1, bits, False, n)[0]
        ntt = hx.NTT(n, q)
        hx.fill_splitmix(x, n, b, 1, q)

        def step():
            ntt.ComputeForward(x, x, 1, 1)
            ntt.ComputeInverse(x, x, 1, 1)
        for _ in range(12):  # the first passes over a fresh buffer run slower
            step()
        sec = event_timed(torch, step, 10)
        other[label] = {"q": q, "ms_per_step": sec * 1e3, "NTT_per_s": 2 * b / sec}
    out["headline_shape_other_moduli"] = other
    del x
    # configs[3] on ONE GPU: the 8 RNS primes x 4096 polynomials (16 GiB) through the
    # multi-modulus entry point (one launch sequence over all primes)
    primes = [18014398510661633, 18014398512365569, 18014398514200577, 18014398514987009,
              18014398515511297, 18014398516559873, 18014398521016321, 18014398524424193]
    b = 4096
    plans = [hx.NTT(n, p) for p in primes]
    x = torch.empty((len(primes), b, n), dtype=torch.int64, device="cuda")
    for k, p in enumerate(primes):
        hx.fill_splitmix(x[k], n, b, 1 + k * b, p)

    def rns_step():
        hx.ComputeForwardRNS(plans, x, x, 1, 1)
        hx.ComputeInverseRNS(plans, x, x, 1, 1)
    for _ in range(6):
        rns_step()
    sec = event_timed(torch, rns_step, 5)
    out["config4_on_one_gpu"] = {
        "shape": f"N={n}, 8 primes (55-bit) x {b} polynomials, hexl_amd_ntt_forward_rns/_inverse_rns",
        "ms_per_step": sec * 1e3, "NTT_per_s": 2 * len(primes) * b / sec}
    return out


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(gpus, argv, port=None):
    """`python bench.py --gpus N` outside torchrun: the command that re-runs this script as
    N ranks of one node (one process per GPU; rendezvous on 127.0.0.1, as the driver does)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
            f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-secondary", action="store_true", help=argparse.SUPPRESS)  # profile runs
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: spawn the ranks ourselves; rank 0's JSON line passes through
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(launcher_command(args.gpus, sys.argv[1:]), env=env))

    import torch

    import hexl_amd as hx

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Dry-run hooks for a 1-GPU box (tests of the multi-rank code path): BENCH_ONE_DEVICE=1
    # puts every rank on cuda:0, BENCH_BACKEND=gloo replaces RCCL for the barrier / max.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if os.environ.get("BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from hexl_amd.sharding import max_over_ranks, shard_range, units_by_prime

    # weak scaling: `world` RNS primes x `batch` polynomials; the flat (prime, poly)
    # unit range is cut into contiguous per-rank shards -> rank g owns prime g
    batch = args.batch
    begin, end = shard_range(world * batch, world, rank)
    (prime_idx, first_poly, count), = units_by_prime(begin, end, batch)
    assert (first_poly, count) == (0, batch)
    q = PRIMES[prime_idx % len(PRIMES)]
    ntt = hx.NTT(N, q)
    data = torch.empty((batch, N), dtype=torch.int64, device="cuda")
    hx.fill_splitmix(data, N, batch, 1 + rank * batch, q)
    check = data[:2].clone()

    def step():
        ntt.ComputeForward(data, data, 1, 1)
        ntt.ComputeInverse(data, data, 1, 1)

    # The first ~10 passes over a freshly allocated 2 GiB buffer run 8 % slower (clock ramp,
    # first-touch page mapping), whatever W is; PREWARM untimed passes precede the W warm-up
    # steps so that short runs measure the steady state too (reported as "prewarm_steps").
    for _ in range(PREWARM + args.warmup):
        step()
    barrier()
    hx.profile_start(8 * args.steps + 16)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    records = hx.profile_stop()
    # fwd followed by inv is the identity: the data must be back where it started
    assert torch.equal(check, data[:2]), "round trip mismatch inside the timed region"

    elapsed = max_over_ranks(elapsed, dist, device="cuda" if backend == "nccl" else "cpu")

    ntts = 2 * batch * args.steps * world
    value = ntts / elapsed
    kern = {}
    for name, ms in records:
        kern.setdefault(name, []).append(ms)
    kern_avg = {k: sum(v) / len(v) for k, v in kern.items()}
    dominant = max(kern_avg, key=lambda k: kern_avg[k] * len(kern[k]))
    alg_bytes = 16.0 * N * batch  # this kernel reads and writes every polynomial once
    achieved = alg_bytes / (kern_avg[dominant] * 1e-3) / 1e9

    # HBM bytes per launch of the dominant kernel from the committed PMC profile
    # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 x2 read
    # correction: profiles/r1_pmc_summary.md); null if the profile does not cover it
    traffic, traffic_source = None, None
    try:
        prof = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)))
        if batch == BATCH:
            traffic = prof["by_bench_kernel_family"].get(dominant)
            traffic_source = ("committed rocprofv3 --pmc profile of this command, not collected in "
                              "this run: profiles/" + TRAFFIC_PROFILE)
    except (OSError, KeyError, ValueError):
        pass

    # Secondary figures (outside the timed region, rank 0 at N=1 only): EltwiseMultMod over the
    # headline batch and BASELINE.json configs[1] / configs[4], each as algorithmic GB/s and
    # fraction of the 8 TB/s HBM peak (algorithmic bytes: SURVEY.md 8d / BASELINE.md 4).
    mult = None
    secondary = None
    if rank == 0 and batch == BATCH and not args.no_secondary:
        mult = timed_eltwise(hx, torch, "EltwiseMultMod(input_mod_factor=1)", N, batch, q)
    if rank == 0 and world == 1 and batch == BATCH and not args.no_secondary:
        del data
        torch.cuda.empty_cache()
        secondary = secondary_configs(hx, torch)
    if rank == 0:
        out = {
            "metric": "Fwd+Inv NTTs/sec, N=65536 q~55b batch=4096",
            "value": value, "unit": "NTT/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "prewarm_steps": PREWARM,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {
                "workload": ("in-place ForwardNTT(1,1) + InverseNTT(1,1), N=65536, "
                             f"55-bit prime per GPU, batch={batch} polys per GPU resident in HBM"),
                "N": N, "batch_per_gpu": batch, "modulus_bits": 55,
                "primes": PRIMES[:max(1, min(world, 8))],
                "parallelism": f"batch-sharded x{world}, no collectives"},
            "hbm_algorithmic_GBps": value * 16.0 * N / 1e9,
            "roofline": {
                "bound": "hbm", "kernel": dominant, "achieved": achieved,
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_kernel_ms": kern_avg,
                "per_kernel": {k: {"ms": v, "GBps_algorithmic": alg_bytes / (v * 1e-3) / 1e9,
                                   "frac_of_hbm_peak": alg_bytes / (v * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                   "limiter": KERNEL_LIMITER.get(k, "hbm")}
                               for k, v in kern_avg.items()},
                "note": ("achieved = algorithmic bytes (16*N per transform, read + write once) of the "
                         "dominant kernel's launch / its HIP-event duration on the launch stream inside "
                         "the timed region; the bound of the path is HBM, `limiter` says what holds each "
                         "kernel below it (the tile pass is VALU-issue limited: DESIGN.md 4-5)")},
        }
        if mult is not None:
            out["eltwise_mult_mod"] = mult
        if secondary is not None:
            out["secondary"] = secondary
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
        elif world > 1:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
