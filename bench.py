"""bench.py -- headline benchmark of the MI355X-native HEXL hot path.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input: an
in-place forward NTT (input_mod_factor = output_mod_factor = 1) followed by an
in-place inverse NTT over 4096 polynomials of degree N = 65536 with a 55-bit
prime (BASELINE.json configs[2]).  Inputs are generated on the device
(splitmix64) and are resident in HBM before the timed region starts.

Multi-GPU (--gpus G > 1, one process per GPU, no data-path collective):
  --scaling auto (default)  G = 1: the headline configuration.  G > 1: `value` is the STRONG job
                            below -- the one BASELINE.json names for the 1/2/4/8 curve -- and the
                            line's "weak" block carries the weak figure measured in the same run
                            (the headline configuration on every GPU: comparable point to point
                            with the one-GPU line; the one-GPU line's secondary.config4 is the
                            strong job's G = 1 point);
  --scaling strong          the job is ALWAYS BASELINE configs[3] -- 8 RNS primes x 4096
                            polynomials = 32,768 transforms per direction -- cut into G
                            contiguous shards of the flat (prime, polynomial) index (SURVEY.md
                            8e; hexl/experimental/seal/key-switch-internal.cpp:51-55 is the
                            per-modulus loop being sharded): one prime per GPU at G = 8, four
                            at G = 2, all eight (16 GiB) at G = 1, each rank running its
                            primes through hexl_amd_ntt_forward_rns / _inverse_rns;
  --scaling weak            rank g transforms the 4096 polynomials of RNS prime g: per-GPU
                            work fixed, the job grows with G.
  (with G > 1 the mode that is not `value` is timed the same way and reported beside it)

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
HBM_ACHIEVABLE_GBPS = 6290.0  # same guide: 6.29 TB/s measured with a float4 copy ("~6.3 achievable")
# Counter-derived figures of the bench kernels (HBM bytes per launch, VALU busy, instructions
# per wave, shader clock): collected with rocprofv3 --pmc by tools/collect_profiles.sh, written
# by tools/summarize_profiles.py together with a hash of the kernel sources they were measured
# on.  They are NOT collected in this run; a profile of other sources is not reported.
COUNTER_PROFILE = "r6_counters.json"
# ... and of every kernel outside the headline configuration (tools/collect_pmc_cells.sh,
# tools/summarize_pmc_cells.py): the one-kernel plans, BASELINE configs[1] / configs[4], the composites
CELL_PROFILE = "r6_pmc_cells.json"
KERNEL_SOURCES = ("ntt_kernels.hip", "modarith.h", "lazy_inverse.h", "tile_geometry.h", "internal.h")
CELL_SOURCES = KERNEL_SOURCES + ("eltwise_kernels.hip", "keyswitch_kernels.hip")
# what keeps each kernel family below the HBM roofline (profiles/r5_pmc_summary.md)
KERNEL_LIMITER = {"ntt_fwd_strided_pass": "hbm", "ntt_inv_strided_pass": "hbm",
                  "ntt_fwd_tile_pass_bottom": "valu-issue + latency",
                  "ntt_inv_tile_pass_bottom": "valu-issue + latency"}
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


HOST_PATH_SHAPES = ((4096, 49), (16384, 54), (65536, 54))  # (N, GeneratePrimes bit size)


def cpu_per_call_baseline(hx):
    """cpu_baseline leg of `host_path`: the oracle's restatement of the reference's algorithm
    (its AVX-512 variant where the host has AVX-512) on ONE thread, one forward transform of one
    polynomial per call, same N and q as host_path -- the figure an unmodified single-call
    caller of the reference sees on this box's CPU."""
    import ctypes as C

    from oracle import hexl_oracle as ho
    simd = bool(ho.lib.ho_has_avx512())
    fwd = ho.lib.ho_ntt_forward_batch_avx512 if simd else ho.lib.ho_ntt_forward_batch
    out = {}
    for n, bits in HOST_PATH_SHAPES:
        q = hx.GeneratePrimes(1, bits, True, n)[0]
        plan = ho.lib.ho_ntt_create(n, q, 0)
        buf = ho.fill_splitmix(n, 11, q)
        p = buf.ctypes.data_as(C.POINTER(C.c_uint64))
        reps = 2000 if n <= 16384 else 400
        for _ in range(20):
            fwd(plan, p, p, 1, 1, 1)
        t0 = time.perf_counter()
        for _ in range(reps):
            fwd(plan, p, p, 1, 1, 1)
        out[f"N={n}"] = {"us_per_call": (time.perf_counter() - t0) / reps * 1e6, "threads": 1,
                         "isa": "avx512 (8 lanes)" if simd else "scalar", "kind": "port"}
        ho.lib.ho_ntt_destroy(plan)
    return out


def kernel_source_hash(sources=None):
    """sha256 over the sources the NTT kernels are compiled from (ties a counter profile to a build)"""
    import hashlib
    h = hashlib.sha256()
    for name in sources or KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "hexl_amd", "csrc", name), "rb").read())
    return h.hexdigest()[:16]


def cell_source_hash():
    return kernel_source_hash(CELL_SOURCES)


def cell_counters():
    """(lookup, note): lookup(cell substring, kernel substring) -> the counter row of that kernel in
    profiles/CELL_PROFILE -- traced duration, fraction of the 8 TB/s peak on algorithmic bytes, HBM
    traffic / algorithmic bytes, VALU busy, waves per SIMD, LDS bank-conflict share -- if the profile
    was collected on the kernel sources of this checkout; otherwise every lookup is None."""
    path = os.path.join(ROOT, "profiles", CELL_PROFILE)
    try:
        prof = json.load(open(path))
    except (OSError, ValueError):
        return (lambda *a: None), "no cell profile committed (profiles/" + CELL_PROFILE + ")"
    if prof.get("kernel_source_sha16") != cell_source_hash():
        return (lambda *a: None), ("profiles/" + CELL_PROFILE + " was collected on other kernel sources ("
                                   + str(prof.get("kernel_source_sha16")) + "): not reported")
    keep = ("kernel", "grid", "traced_us", "frac_of_peak", "traffic_ratio", "valu_busy", "waves_per_simd",
            "wait_any_frac", "lds_conflict_share", "lds_busy", "valu_per_wave", "clock_GHz")

    def lookup(cell, kernel):
        for r in prof.get("rows", []):
            if cell in (r.get("cell") or "") and kernel in r["kernel"]:
                return {k: r.get(k) for k in keep}
        return None
    return lookup, ("committed rocprofv3 --pmc profile (tools/collect_pmc_cells.sh) on these kernel sources (sha16 "
                    + prof["kernel_source_sha16"] + "), not collected in this run: profiles/" + CELL_PROFILE)


def counter_profile():
    """(per-kernel-family counters, note) from profiles/COUNTER_PROFILE if it was measured on
    the kernel sources of this checkout, else ({}, why not)"""
    path = os.path.join(ROOT, "profiles", COUNTER_PROFILE)
    try:
        prof = json.load(open(path))
    except (OSError, ValueError):
        return {}, "no counter profile committed (profiles/" + COUNTER_PROFILE + ")"
    if prof.get("kernel_source_sha16") != kernel_source_hash():
        return {}, ("profiles/" + COUNTER_PROFILE + " was collected on other kernel sources ("
                    + str(prof.get("kernel_source_sha16")) + "): not reported")
    return prof.get("by_bench_kernel_family", {}), (
        "committed rocprofv3 --pmc profile of this command on these kernel sources (sha16 "
        + prof["kernel_source_sha16"] + "), not collected in this run: profiles/" + COUNTER_PROFILE)


def median(v):
    v = sorted(v)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


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


def graph_timed(torch, fn, calls, replays=40):
    """average seconds per call of fn replayed from a captured HIP graph of `calls` calls"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            for _ in range(calls):
                fn()
    torch.cuda.synchronize()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (replays * calls) * 1e-3


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
        "timing": "eager: calls issued back to back on one stream (launch gaps included)",
        "fwd": rate(16.0 * n * b, event_timed(torch, lambda: ntt.ComputeForward(y, x, 1, 1), 200)),
        "inv": rate(16.0 * n * b, event_timed(torch, lambda: ntt.ComputeInverse(y, x, 1, 1), 200)),
        "multmod": rate(24.0 * n * b,
                        event_timed(torch, lambda: hx.EltwiseMultMod(y, x, x, n * b, q, 1), 200)),
    }
    # the same calls from C++ (tests/cpp/eager_rate.cpp): no interpreter between the launches
    out["config2"]["from_cpp"] = guarded("eager_rate", lambda: run_eager_rate(n, b))
    # the same calls replayed from a captured HIP graph of 32 back-to-back launches: what a
    # caller that batches its launches sees (the C-ABI launches are stream-ordered and
    # allocation-free, hence capturable); benchmark/bench-ntt.cpp:212-239 is the reference's
    # own N = 4096 loop
    try:
        out["config2"]["graph_of_32"] = {
            "fwd": rate(16.0 * n * b, graph_timed(torch, lambda: ntt.ComputeForward(y, x, 1, 1), 32)),
            "inv": rate(16.0 * n * b, graph_timed(torch, lambda: ntt.ComputeInverse(y, x, 1, 1), 32)),
            "multmod": rate(24.0 * n * b,
                            graph_timed(torch, lambda: hx.EltwiseMultMod(y, x, x, n * b, q, 1), 32)),
        }
    except Exception as e:  # noqa: BLE001 -- graph capture is a measurement aid, not the product
        out["config2"]["graph_of_32"] = {"error": f"{type(e).__name__}: {e}"[:200]}
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
                        ("57-bit prime (Lazy32 policy)", 56), ("59-bit prime (Lazy16 policy)", 58),
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
        hx.profile_start(64)
        for _ in range(4):
            step()
        kern = {}
        for name, ms in hx.profile_stop():
            kern.setdefault(name, []).append(ms)
        other[label] = {"q": q, "ms_per_step": sec * 1e3, "NTT_per_s": 2 * b / sec,
                        "avg_kernel_ms": {k: sum(v) / len(v) for k, v in kern.items()},
                        # algorithmic bytes of the step's two transforms / its time / 8 TB/s
                        "transform_frac": 2 * 16.0 * n * b / sec / 1e9 / HBM_PEAK_GBPS}
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
    # the committed counter profile of these kernels (same shapes, same kernel sources)
    look, note = cell_counters()
    out["counters_source"] = note
    c2 = "N=2^12 49-bit (Fp64) x 256"
    out["config2"]["counters"] = {
        "fwd": look(c2, "tile_pass<true, 12"), "inv": look(c2, "tile_pass<false, 12"),
        "multmod": look("EltwiseMultMod configs[1]", "eltwise_vec2<MultOp"),
        "at_1_GiB_batches": {"fwd": look("N=2^12 49-bit (Fp64) x 32768", "tile_pass<true, 12"),
                             "inv": look("N=2^12 49-bit (Fp64) x 32768", "tile_pass<false, 12")}}
    out["config5"]["counters"] = {
        "fma": look("EltwiseFMAMod configs[4]", "FmaOp"), "reduce": look("EltwiseReduceMod configs[4]", "ReduceOp"),
        "fused": look("fused ReduceMod+FMAMod", "ReduceFmaOp")}
    out["one_kernel_plans_counters"] = {
        f"N=2^{ln} {bits}-bit ({pol})": {"fwd": look(f"N=2^{ln} {bits}-bit ({pol})", "<true, "),
                                         "inv": look(f"N=2^{ln} {bits}-bit ({pol})", "<false, ")}
        for ln in (13, 14) for bits, pol in ((28, "Small"), (44, "Fp64L"), (49, "Fp64"), (55, "Lazy"), (60, "Harvey60"))}
    for label, cell in (("30-bit prime (Small policy)", "N=2^16 28-bit (Small)"),
                        ("50-bit prime (Fp64 policy)", "N=2^16 49-bit (Fp64)")):
        if label in other:
            other[label]["counters"] = {
                "note": "cell profile collected on the first prime GeneratePrimes(1, b, true, N) returns: same policy",
                "fwd_strided": look(cell, "strided_pass<true"), "fwd_tile": look(cell, "tile_pass<true, 11"),
                "inv_tile": look(cell, "tile_pass<false, 11"), "inv_strided": look(cell, "strided_pass<false")}
    return out


def sustained_run(torch, step, polys, seconds=3.0):
    """The same step loop for >= `seconds`: the power-capped steady state (a burst from idle
    runs up to 15 % off it) and long enough for an outside sampler to see the
    GPU busy.  Every step is bracketed by HIP events on the launch stream; queued 64 at a time."""
    if seconds <= 0:  # (profile runs: BENCH_SUSTAINED_S=0)
        return None
    torch.cuda.synchronize()
    ms, t0 = [], time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(65)]
        marks[0].record()
        for i in range(64):
            step()
            marks[i + 1].record()
        torch.cuda.synchronize()
        ms += [marks[i].elapsed_time(marks[i + 1]) for i in range(64)]
    wall = time.perf_counter() - t0
    srt = sorted(ms)
    med = median(ms)
    return {"seconds": wall, "steps": len(ms), "ms_per_step_median": med,
            "ms_per_step_mean_wall": wall / len(ms) * 1e3,
            "ms_per_step_p10": srt[len(srt) // 10], "ms_per_step_p90": srt[(len(srt) * 9) // 10],
            "NTT_per_s": 2 * polys / (med * 1e-3),
            "NTT_per_s_wall": 2 * polys * len(ms) / wall}


def host_path(hx):
    """What an UNMODIFIED caller of the reference gets: intel::hexl::NTT::ComputeForward on ONE
    polynomial in host memory per call (hexl/include/hexl/ntt/ntt.hpp:99-110), synchronous --
    through hexl_amd_ntt_forward_host, the entry point the C++ shim binds.  Per call, by kind of
    buffer: ordinary (pageable) host memory; pinned device-mapped host memory
    (hexl_amd_host_alloc = intel::hexl::DeviceMappedAllocator); device memory + a
    synchronisation; device memory, calls queued.  Beside it the CPU: the oracle's AVX-512
    variant of the reference's algorithm on ONE host thread, same N and q (the reference itself
    is single-threaded per call)."""
    import ctypes as C

    import numpy as np
    import torch

    out = {"unit": "us per call, one forward transform of one polynomial",
           "entry_point": "hexl_amd_ntt_forward_host (what intel::hexl::NTT::ComputeForward binds)",
           "measured_through": "Python ctypes around the C-ABI (adds about a microsecond of "
                               "call overhead per call); cpp_budget is the same call made from C++"}
    # the same call from C++ (intel::hexl::NTT::ComputeForward on a std::vector, clock_gettime
    # around it) with its phases timed one by one: tests/cpp/host_call_budget.cpp
    if os.path.exists(HOST_CALL_BUDGET_BIN):
        import subprocess
        r = subprocess.run([HOST_CALL_BUDGET_BIN, "1000"], capture_output=True, text=True, timeout=300)
        rows = [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]
        out["cpp_budget"] = ({f"N={row['n']}": row for row in rows} if r.returncode == 0 and rows
                             else {"error": (r.stderr or r.stdout)[-300:]})
    for n, bits in HOST_PATH_SHAPES:
        q = hx.GeneratePrimes(1, bits, True, n)[0]
        ntt = hx.NTT(n, q)
        src = np.random.default_rng(11).integers(0, q, n, dtype=np.uint64)
        reps = 300 if n <= 16384 else 150
        row = {"q": q}

        def timed(call, sync_each=False):
            for _ in range(20):
                call()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                call()
                if sync_each:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e6
        a, b = src.copy(), np.zeros(n, dtype=np.uint64)
        pa, pb = a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)
        row["ordinary_host_memory"] = timed(lambda: hx.lib.hexl_amd_ntt_forward_host(ntt._h, pb, pa, 1, 1, 1))
        want = b.copy()
        pm = C.c_void_p()
        if hx.lib.hexl_amd_host_alloc(C.byref(pm), 2 * n * 8) == 0:
            m = np.ctypeslib.as_array(C.cast(pm, C.POINTER(C.c_uint64)), shape=(2 * n,))
            m[:n] = src
            po = C.c_void_p(pm.value + n * 8)
            row["device_mapped_host_memory"] = timed(
                lambda: hx.lib.hexl_amd_ntt_forward_host(ntt._h, po, pm, 1, 1, 1))
            row["results_identical"] = bool(np.array_equal(m[n:], want))
            del m
            hx.lib.hexl_amd_host_free(pm)
        d = hx.from_numpy(src)
        o = torch.empty_like(d)
        row["device_memory_plus_sync"] = timed(lambda: ntt.ComputeForward(o, d, 1, 1), sync_each=True)
        row["device_memory_queued"] = timed(lambda: ntt.ComputeForward(o, d, 1, 1))
        out[f"N={n}"] = row
    return out


def composites(hx):
    """The in-tree composite callers on device buffers: KeySwitch at a
    CKKS-like shape, per target, one target per call and 256 per call; DyadicMultiply as GB/s at
    56 bytes per coefficient (2 x 2 polynomials in, 3 out)."""
    import numpy as np
    import torch
    rng = np.random.default_rng(1)

    def gpu_time(fn, reps):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    out = {}
    n, k = 32768, 16
    moduli = hx.GeneratePrimes(k, 54, True, n)
    x = hx.from_numpy(np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli] * 2))
    y = hx.from_numpy(np.concatenate([rng.integers(0, q, n, dtype=np.uint64) for q in moduli] * 2))
    r = torch.empty(3 * n * k, dtype=torch.int64, device="cuda")
    t = gpu_time(lambda: hx.DyadicMultiply(r, x, y, n, moduli), 50)
    out["dyadic_multiply"] = dict(rate(56.0 * n * k, t), shape=f"n={n} x {k} moduli (55-bit), one pair per call",
                                  bytes_per_coefficient=56)
    pairs = 64
    xb, yb = x.repeat(pairs), y.repeat(pairs)
    rb = torch.empty(3 * n * k * pairs, dtype=torch.int64, device="cuda")
    t = gpu_time(lambda: hx.DyadicMultiplyBatch(rb, xb, yb, pairs, n, moduli), 10)
    out["dyadic_multiply_batch"] = dict(rate(56.0 * n * k * pairs, t),
                                        shape=f"n={n} x {k} moduli, {pairs} pairs per call")
    del x, y, r, xb, yb, rb
    n, D, C = 16384, 7, 2
    K = D + 1
    moduli = hx.GeneratePrimes(K, 54, True, n)
    keys = [hx.from_numpy(np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                                          for _ in range(C) for i in range(K)])) for _ in range(D)]
    msf = [int(rng.integers(1, moduli[i], dtype=np.uint64)) for i in range(D)]
    target = np.concatenate([rng.integers(0, moduli[j], n, dtype=np.uint64) for j in range(D)])
    result = np.concatenate([rng.integers(0, moduli[i], n, dtype=np.uint64)
                             for _ in range(C) for i in range(D)])
    ks = {"shape": f"n={n}, {D} decomposition moduli (55-bit) + special prime, {C} key components"}
    d_t, d_r = hx.from_numpy(target), hx.from_numpy(result)
    # one target per call, as the reference's caller makes it (key-switch-internal.cpp:25-201):
    # on the default stream (always launch by launch) and on a stream of the caller's own, where a
    # sequence that comes back with the same buffers is replayed from a captured graph
    # (include/hexl_amd.h: "ks_graph"); the counters say which path the timed calls took
    t = gpu_time(lambda: hx.KeySwitch(d_r, d_t, n, D, K, D + 1, C, moduli, keys, msf), 30)
    ks["one_target_per_call_default_stream_us"] = t * 1e6
    side = torch.cuda.Stream()

    def on_side():
        with torch.cuda.stream(side):
            hx.KeySwitch(d_r, d_t, n, D, K, D + 1, C, moduli, keys, msf)

    def side_time(reps):
        for _ in range(4):
            on_side()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0 = time.perf_counter()
        with torch.cuda.stream(side):
            e0.record()
        for _ in range(reps):
            on_side()
        with torch.cuda.stream(side):
            e1.record()
        enqueue = time.perf_counter() - w0
        side.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3, enqueue / reps
    hx.set_tuning("ks_graph", 0)
    t_eager, host_eager = side_time(30)
    hx.set_tuning("ks_graph", 1)
    r0 = hx.get_counter("ks_graph_replays")
    t_replay, host_replay = side_time(30)
    replays = hx.get_counter("ks_graph_replays") - r0
    ks["one_target_per_call_eager_us"] = t_eager * 1e6
    ks["one_target_per_call_us"] = t_replay * 1e6
    ks["one_target_per_call_path"] = (f"graph replay ({replays} of 34 calls)" if replays >= 30
                                      else f"launch by launch ({replays} replays)")
    ks["host_enqueue_us_per_call"] = {"eager": host_eager * 1e6, "replay": host_replay * 1e6}
    # the same call from C++ (tests/cpp/ks_call_cost.cpp: no interpreter between the calls)
    ks["from_cpp"] = guarded("ks_call_cost", lambda: run_ks_call_cost(n, D))
    hx.lib.hexl_amd_release_stream_workspaces(side.cuda_stream)
    # Floors for one target: (a) its transforms alone at the batched per-transform rate of this
    # degree -- D inverse (targets to coefficient form) + D^2 forward (operands) + C inverse (last
    # components) + C D forward (corrections); (b) its algorithmic HBM bytes at 8 TB/s -- target
    # D n, key blocks (D + 1) D C n, result read + written C D n each, 8 bytes a word.
    ref_q = moduli[0]
    plan = hx.NTT(n, ref_q)
    polys = 4096
    buf = torch.empty((polys, n), dtype=torch.int64, device="cuda")
    hx.fill_splitmix(buf, n, polys, 7, ref_q)
    t_fwd = gpu_time(lambda: plan.ComputeForward(buf, buf, 1, 1), 10) / polys
    t_inv = gpu_time(lambda: plan.ComputeInverse(buf, buf, 1, 1), 10) / polys
    del buf
    n_fwd, n_inv = D * D + C * D, D + C
    ntt_floor = n_fwd * t_fwd + n_inv * t_inv
    hbm_bytes = 8.0 * n * (D + (D + 1) * D * C + 2 * C * D)
    ks["transforms_per_target"] = {"forward": n_fwd, "inverse": n_inv,
                                   "ns_per_transform_batched": {"forward": t_fwd * 1e9, "inverse": t_inv * 1e9}}
    ks["ntt_floor_us"] = ntt_floor * 1e6
    ks["hbm_floor_us"] = hbm_bytes / (HBM_PEAK_GBPS * 1e9) * 1e6
    ks["algorithmic_bytes_per_target"] = hbm_bytes
    for T in (256,):
        d_tt, d_rr = hx.from_numpy(np.tile(target, T)), hx.from_numpy(np.tile(result, T))
        t = gpu_time(lambda: hx.KeySwitchBatch(d_rr, d_tt, T, n, D, K, D + 1, C, moduli, keys, msf), 5)
        ks[f"{T}_targets_per_call_us_per_target"] = t * 1e6 / T
        ks[f"{T}_targets_per_call_ms"] = t * 1e3
        ks[f"{T}_targets_frac_of_ntt_floor"] = ntt_floor / (t / T)
        ks[f"{T}_targets_GBps_algorithmic"] = hbm_bytes * T / t / 1e9
        ks[f"{T}_targets_frac_of_hbm_peak"] = hbm_bytes * T / t / 1e9 / HBM_PEAK_GBPS
        del d_tt, d_rr
    ks["one_target_frac_of_ntt_floor"] = ntt_floor / t_replay
    ks["one_target_frac_of_ntt_floor_eager"] = ntt_floor / t_eager
    # the committed counter profile of the 256-target call's kernels (same shape, same sources)
    look, note = cell_counters()
    cell = "KeySwitchBatch 256 targets n=16384"
    ks["counters_256_targets"] = {
        "source": note,
        "targets_inverse": look(cell, "tile_pass_multi<false, 14"),
        "operands_forward": look(cell, "false, false>"),
        "multiply_accumulate": look(cell, "ks_mac_kernel"),
        "fused_tail_round_forward_finish": look("fused tail", "false, true>"),
        "launches_per_call": "5 (one-kernel transforms): inverse of the targets, forward of the operands through a "
                             "source map, multiply-accumulate, inverse of the last components, fused tail"}
    out["key_switch"] = ks
    torch.cuda.empty_cache()
    return out


HOST_CALL_BUDGET_BIN = os.path.join(ROOT, "tests", "cpp", "host_call_budget")
KS_CALL_COST_BIN = os.path.join(ROOT, "tests", "cpp", "ks_call_cost")


EAGER_RATE_BIN = os.path.join(ROOT, "tests", "cpp", "eager_rate")


def run_eager_rate(n, b):
    import subprocess
    if not os.path.exists(EAGER_RATE_BIN):
        raise SystemExit(f"{EAGER_RATE_BIN} is missing: python -c 'import __graft_entry__ as g; g.build()'")
    r = subprocess.run([EAGER_RATE_BIN], capture_output=True, text=True, timeout=120)
    rows = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{")]
    if r.returncode != 0 or not rows or rows[0]["n"] != n or rows[0]["batch"] != b:
        raise SystemExit(f"eager_rate failed (rc {r.returncode}): {r.stderr[-300:]}")
    row = rows[0]
    return {"timing": "calls issued back to back on one stream from C++, wall time per call over 2000 calls",
            "fwd": rate(16.0 * n * b, row["fwd_us"] * 1e-6), "inv": rate(16.0 * n * b, row["inv_us"] * 1e-6),
            "multmod": rate(24.0 * n * b, row["multmod_us"] * 1e-6)}


def run_ks_call_cost(n, D):
    import subprocess
    if not os.path.exists(KS_CALL_COST_BIN):
        raise SystemExit(f"{KS_CALL_COST_BIN} is missing: python -c 'import __graft_entry__ as g; g.build()'")
    r = subprocess.run([KS_CALL_COST_BIN, str(n), str(D)], capture_output=True, text=True, timeout=120)
    rows = [json.loads(line) for line in r.stdout.splitlines() if line.startswith("{")]
    if r.returncode != 0 or not rows:
        raise SystemExit(f"ks_call_cost failed (rc {r.returncode}): {r.stderr[-300:]}")
    return {("replayed" if row["ks_graph"] else "launch_by_launch") + (", walking over 8 ciphertexts"
            if row["buffers"].startswith("walking") else ", one ciphertext"):
            {"wall_us_per_call": row["wall_us_per_call"], "host_enqueue_us": row["host_enqueue_us"],
             "call_and_wait_us": row["call_and_wait_us"]} for row in rows}


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


def guarded(name, fn):
    """fn(), or {"error": ...} in its place: for the blocks reported beside the headline."""
    try:
        return fn()
    except (Exception, SystemExit) as e:  # noqa: BLE001
        sys.stderr.write(f"bench.py: {name} failed: {type(e).__name__}: {e}\n")
        return {"error": f"{type(e).__name__}: {e}"[:300]}


MULTI_DEVICE_BIN = os.path.join(ROOT, "tests", "cpp", "multi_device")


def run_multi_device(devices, scaling, steps, warmup, batch=BATCH, n=N, timeout=900):
    """tests/cpp/multi_device: ONE process, one std::thread + stream + plans per GPU over the
    C-ABI alone (what a C++ caller of the reference does).  Returns its JSON."""
    import subprocess
    if not os.path.exists(MULTI_DEVICE_BIN):
        raise SystemExit(f"{MULTI_DEVICE_BIN} is missing: python -c 'import __graft_entry__ as g; g.build()'")
    cmd = [MULTI_DEVICE_BIN, "--devices", ",".join(str(d) for d in devices), "--scaling", scaling,
           "--n", str(n), "--batch", str(batch), "--primes", str(len(PRIMES)), "--bits", "54",
           "--steps", str(steps), "--warmup", str(warmup)]
    if n == N:  # the per-worker probe: the definition digests of the primes of this job (committed data)
        for c in json.load(open(os.path.join(ROOT, "tests", "golden", "ntt_definition_fixtures.json")))["cases"]:
            if c["n"] == N and c["q"] in PRIMES:
                cmd += ["--probe", f"{c['q']}:{c['forward']['sha256_le_u64']}:{c['inverse']['sha256_le_u64']}"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    try:
        out = json.loads(line)
    except ValueError:
        raise SystemExit(f"multi_device failed (rc {r.returncode}): {r.stdout[-500:]} {r.stderr[-500:]}")
    if not out.get("ok"):
        raise SystemExit(f"multi_device: {out.get('error')}")
    return out


def threads_main(args):
    """--launcher threads: the same job as the one-process-per-GPU launcher, driven from ONE
    process through the C-ABI (tests/cpp/multi_device.cpp), same JSON shape."""
    devices = [0] * args.gpus if os.environ.get("BENCH_ONE_DEVICE") == "1" else list(range(args.gpus))
    if args.scaling == "auto":
        args.scaling = "strong" if args.gpus > 1 else "weak"
    md = run_multi_device(devices, args.scaling, args.steps, args.warmup + PREWARM // 2, batch=args.batch)
    strong = args.scaling == "strong"
    total = md["polynomials_total"]
    print(json.dumps({
        "metric": "Fwd+Inv NTTs/sec, N=65536 q~55b batch=4096",
        "value": md["value"], "unit": "NTT/s", "n_gpus": args.gpus, "steps": args.steps,
        "warmup": args.warmup, "prewarm_steps": PREWARM // 2, "ms_per_step": md["ms_per_step"],
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "u64",
        "data": "synthetic", "per_rank_NTT_per_s": md["per_rank_NTT_per_s"],
        "per_rank_probe_ok": md.get("per_rank_probe_ok"), "per_rank_plan_device": md.get("per_rank_plan_device"),
        "launcher": "threads", "rendezvous": "none", "rendezvous_note": "one process: std::thread per GPU",
        "devices": md["devices"], "visible_devices": md["visible_devices"],
        "verified": {"probe_polynomials_compared": md["probe_polynomials_compared"],
                     "probe_mismatches": md["probe_mismatches"], "ref_device": md["ref_device"]},
        "config": {
            "workload": (("BASELINE configs[3]: in-place ForwardNTT(1,1) + InverseNTT(1,1), N=65536, "
                          f"8 RNS primes (55-bit) x {args.batch} polynomials = {total} transforms per "
                          f"direction, sharded over {args.gpus} GPU(s), resident in HBM") if strong else
                         ("in-place ForwardNTT(1,1) + InverseNTT(1,1), N=65536, "
                          f"55-bit prime per GPU, batch={args.batch} polys per GPU resident in HBM")),
            "N": N, "batch_per_gpu": md["per_rank_polynomials"][0], "modulus_bits": 55,
            "polynomials_total": total,
            "parallelism": (f"{args.scaling} scaling: flat (prime, polynomial) index sharded x{args.gpus}, "
                            "one host thread + stream + plans per GPU in one process (C-ABI), no collectives")},
        "hbm_algorithmic_GBps": md["value"] * 16.0 * N / 1e9,
        "roofline": None, "cpu_baseline": None,
    }))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=("auto", "weak", "strong"),
                    default=os.environ.get("BENCH_SCALING", "auto"),
                    help="multi-GPU job: strong = always BASELINE configs[3], 8 primes x 4096 "
                         "polynomials sharded over the GPUs; weak = one prime x 4096 polynomials per "
                         "GPU; auto (default) = the headline configuration on one GPU, strong on "
                         "several with the weak figure in the line's \"weak\" block")
    ap.add_argument("--launcher", choices=("processes", "threads"),
                    default=os.environ.get("BENCH_LAUNCHER", "processes"),
                    help="processes (default): one process per GPU under torch.distributed.run; "
                         "threads: one process, one std::thread + stream + plans per GPU over the "
                         "C-ABI (tests/cpp/multi_device.cpp)")
    ap.add_argument("--batch", type=int, default=BATCH, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-secondary", action="store_true", help=argparse.SUPPRESS)  # profile runs
    # (profile runs: the probe's one-polynomial launches use the headline's kernels and would enter
    # rocprofv3's per-kernel-name averages -- 1 launch in 25 at 1/4096 of the work: -4 % on every
    # average; tools/collect_profiles.sh passes it, the line then says "per_rank_probe_ok": null)
    ap.add_argument("--no-probe", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.launcher == "threads":
        if int(os.environ.get("RANK", "0")) == 0:  # (under torchrun: rank 0 drives all devices)
            threads_main(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: spawn the ranks ourselves; rank 0's JSON line passes through
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(launcher_command(args.gpus, sys.argv[1:]), env=env))

    # Exactly ONE line on stdout, the JSON: libraries under us write banners to file descriptor 1
    # (gloo's "Rank 0 is connected to ...", RCCL's version block), so everything but the line goes
    # to stderr from here on and the line is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch

    import hexl_amd as hx

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # Dry-run hooks for a 1-GPU box (tests of the multi-rank code path): BENCH_ONE_DEVICE=1
    # puts every rank on cuda:0, BENCH_BACKEND=gloo skips the RCCL probe.
    from hexl_amd.sharding import job_partition, rendezvous
    if os.environ.get("BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    # fewer visible devices than ranks (a launcher that masks devices per rank, or a box with
    # fewer GPUs than --gpus): the ranks share what is there and the line says so -- a shared
    # device is a dry run of the code path, not a scaling measurement
    visible = torch.cuda.device_count()
    oversubscribed = visible > 0 and world > visible
    if visible > 0 and local_rank >= visible:
        local_rank %= visible
    torch.cuda.set_device(local_rank)
    # The ranks meet for a barrier and two scalar reductions only (no data-path collective):
    # gloo is always there, RCCL is used when -- and only when -- every rank brought it up
    # (hexl_amd/sharding.py: rendezvous); the line says which in "rendezvous".
    rv = rendezvous(rank, world, local_rank, prefer=os.environ.get("BENCH_BACKEND", "nccl"))

    def barrier():
        rv.barrier()
        torch.cuda.synchronize()


    # The job: weak = `world` primes x `batch` polynomials (rank g owns prime g); strong =
    # the 8 primes x `batch` polynomials of configs[3] whatever `world` is.  Either way the flat
    # (prime, polynomial) index is cut into contiguous per-rank shards, no collective on the
    # data path.  "auto" (the default, what the driver's launcher gets): one GPU = the headline
    # configuration (configs[2]); several GPUs = configs[3] sharded, the job BASELINE.json names
    # for the 1/2/4/8 curve, with the weak figure (one prime x `batch` per GPU: the headline
    # configuration on every GPU, comparable point to point with the one-GPU line) measured
    # in the same run and reported in the line's "weak" block.
    batch = args.batch
    if args.scaling == "auto":
        args.scaling = "strong" if world > 1 else "weak"
    strong = args.scaling == "strong"

    def build_job(scaling):
        """This rank's shard of the job: (segments, plans, data, check, step, polynomials)."""
        primes = len(PRIMES) if scaling == "strong" else world
        segs = job_partition(primes, batch, world, scaling)[rank]
        polys = sum(c for _, _, c in segs)
        plans_ = [hx.NTT(N, PRIMES[p % len(PRIMES)]) for p, _, _ in segs]
        data_ = torch.empty((polys, N), dtype=torch.int64, device="cuda")
        views_, off = [], 0
        for (p, first, count), plan in zip(segs, plans_):
            v = data_[off:off + count]
            hx.fill_splitmix(v, N, count, 1 + p * batch + first, PRIMES[p % len(PRIMES)])
            views_.append(v)
            off += count
        whole_ = len(segs) > 1 and all(c == segs[0][2] for _, _, c in segs)

        def step_():
            if len(segs) == 1:
                plans_[0].ComputeForward(data_, data_, 1, 1)
                plans_[0].ComputeInverse(data_, data_, 1, 1)
            elif whole_:  # several whole primes: the RNS entry point (prime-major blocks)
                hx.ComputeForwardRNS(plans_, data_, data_, 1, 1)
                hx.ComputeInverseRNS(plans_, data_, data_, 1, 1)
            else:  # a shard that cuts through primes: segment by segment
                for plan, v in zip(plans_, views_):
                    plan.ComputeForward(v, v, 1, 1)
                for plan, v in zip(plans_, views_):
                    plan.ComputeInverse(v, v, 1, 1)
        return segs, plans_, data_, data_[:2].clone(), step_, polys, primes

    def side_job(scaling):
        """The other scaling mode of a multi-GPU run, timed like the headline (K steps between
        barriers, max over ranks, round trip asserted), reported beside it."""
        segs, plans_, data_, check_, step_, polys, primes = build_job(scaling)
        for _ in range(PREWARM + args.warmup):
            step_()
        barrier()
        t_ = time.perf_counter()
        for _ in range(args.steps):
            step_()
        barrier()
        dt = time.perf_counter() - t_
        assert torch.equal(check_, data_[:2]), "round trip mismatch in the side job"
        rate = rv.gather(2 * polys * args.steps / dt)
        dt = rv.max(dt)
        total = primes * batch
        del data_
        torch.cuda.empty_cache()
        return {"scaling": scaling, "value": 2 * total * args.steps / dt, "unit": "NTT/s",
                "ms_per_step": dt / args.steps * 1e3, "steps": args.steps,
                "polynomials_total": total, "polynomials_this_rank": polys,
                "per_rank_NTT_per_s": rate,
                "workload": (f"configs[3]: 8 primes x {batch} polynomials sharded over {world} GPU(s)"
                             if scaling == "strong" else
                             f"the headline configuration on every GPU: one prime x {batch} polynomials each")}

    segments, plans, data, check, step, my_polys, num_primes = build_job(args.scaling)

    # Before anything is timed every rank proves that ITS plans on ITS device compute the right
    # thing: for each prime of its shard the forward transform of splitmix64(seed = 1) mod q and the
    # inverse transform of splitmix64(seed = 1001) mod q, N = 65536, are hashed and compared with the
    # digests of tests/golden/ntt_definition_fixtures.json (committed data, pinned to the big-integer
    # definition of the transform: no oracle in this process).  A round trip alone would also pass
    # with another prime's tables or on another device's memory; this does not.
    def rank_probe():
        import hashlib
        fx = {c["q"]: c for c in json.load(open(os.path.join(
            ROOT, "tests", "golden", "ntt_definition_fixtures.json")))["cases"] if c["n"] == N}
        ok, devices = True, set()
        for (prime, _, _), plan in zip(segments, plans):
            q = PRIMES[prime % len(PRIMES)]
            devices.add(plan.GetDevice())
            for which, seed in (("forward", 1), ("inverse", 1001)):
                v = torch.empty((1, N), dtype=torch.int64, device="cuda")
                hx.fill_splitmix(v, N, 1, seed, q)
                (plan.ComputeForward if which == "forward" else plan.ComputeInverse)(v, v, 1, 1)
                digest = hashlib.sha256(hx.to_numpy(v).astype("<u8").tobytes()).hexdigest()
                ok = ok and q in fx and digest == fx[q][which]["sha256_le_u64"]
        expect = int(os.environ.get("BENCH_PROBE_EXPECT_DEVICE", local_rank))
        return ok and devices == {expect}, sorted(devices)

    if args.no_probe:
        probe_ok, probe_devices = True, sorted({plan.GetDevice() for plan in plans})
    else:
        probe_ok, probe_devices = rank_probe()
    per_rank_probe_ok = [bool(v) for v in rv.gather(1.0 if probe_ok else 0.0)]
    per_rank_device = [int(v) for v in rv.gather(float(probe_devices[0] if probe_devices else -1))]
    if args.no_probe:
        per_rank_probe_ok = None
    elif not all(per_rank_probe_ok):
        raise SystemExit(f"per-rank probe FAILED: ok per rank {per_rank_probe_ok}, plan device per rank "
                         f"{per_rank_device} (rank {rank}: local rank {local_rank}, plans on {probe_devices})")

    # The first ~10 passes over a freshly allocated 2 GiB buffer run 8 % slower (clock ramp,
    # first-touch page mapping), whatever W is; PREWARM untimed passes precede the W warm-up
    # steps so that short runs measure the steady state too (reported as "prewarm_steps").
    for _ in range(PREWARM + args.warmup):
        step()
    barrier()
    hx.profile_start(8 * len(segments) * args.steps + 16)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()  # (on the launch stream; costs no synchronisation)
    barrier()
    elapsed = time.perf_counter() - t0
    records = hx.profile_stop()
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    # fwd followed by inv is the identity: the data must be back where it started
    assert torch.equal(check, data[:2]), "round trip mismatch inside the timed region"

    # the steady state: the same loop for >= 3 s (outside the K timed steps)
    sustained = sustained_run(torch, step, my_polys, float(os.environ.get("BENCH_SUSTAINED_S", "3")))
    my_rate = 2 * my_polys * args.steps / elapsed
    per_rank = rv.gather(my_rate)
    # several GPUs: the other scaling mode in the same run (every rank takes part)
    other_mode = None
    if world > 1 and not args.no_secondary:
        other = "weak" if strong else "strong"
        other_mode = (other, side_job(other))
    median_ms = rv.max(median(step_ms))
    elapsed = rv.max(elapsed)

    total_polys = num_primes * batch
    ntts = 2 * total_polys * args.steps
    value = ntts / elapsed
    kern = {}
    for name, ms in records:
        kern.setdefault(name, []).append(ms)
    kern_avg = {k: sum(v) / len(v) for k, v in kern.items()}
    kern_med = {k: median(v) for k, v in kern.items()}
    dominant = max(kern_avg, key=lambda k: kern_avg[k] * len(kern[k]))
    # a launch of rank 0 covers the polynomials of one of its segments (a shard of several whole
    # primes goes through the RNS entry point, which at this batch size launches prime by prime:
    # checked on the two-rank dry run, 4096 polynomials per launch); each kernel reads and
    # writes every polynomial once
    alg_bytes = 16.0 * N * segments[0][2]
    achieved = alg_bytes / (kern_avg[dominant] * 1e-3) / 1e9
    # a TRANSFORM is two launches (two HBM round trips above N = 2^14): its algorithmic bytes
    # over the time of both kernels -- the figure the per-launch fraction does not show
    fam = {"fwd": ("ntt_fwd_strided_pass", "ntt_fwd_tile_pass_bottom"),
           "inv": ("ntt_inv_tile_pass_bottom", "ntt_inv_strided_pass")}
    transform = {}
    for direction, names in fam.items():
        if all(k in kern_avg for k in names):
            ms = sum(kern_avg[k] for k in names)
            transform[direction] = {"ms": ms, "GBps_algorithmic": alg_bytes / (ms * 1e-3) / 1e9,
                                    "frac_of_hbm_peak": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                    "frac_of_achievable": alg_bytes / (ms * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBPS}
    kernel_ms_per_step = sum(kern_avg[k] * len(kern[k]) for k in kern) / args.steps
    transform_frac = 2 * 16.0 * N * my_polys / (kernel_ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS

    # counter-derived figures per launch of each kernel (HBM bytes: FETCH_SIZE / WRITE_SIZE in
    # separate passes with the gfx950 x2 read correction; VALU busy: SQ_ACTIVE_INST_VALU quad-
    # cycles over SIMD-cycles) from the committed profile of THESE kernel sources, else null
    counters, counters_note = ({}, "profile covers the default batch only")
    if batch == BATCH and segments[0][2] == BATCH:
        counters, counters_note = counter_profile()
    traffic = (counters.get(dominant) or {}).get("hbm_bytes_per_launch")

    # Secondary figures (outside the timed region, rank 0 at N=1 only): EltwiseMultMod over the
    # headline batch and BASELINE.json configs[1] / configs[4], each as algorithmic GB/s and
    # fraction of the 8 TB/s HBM peak (algorithmic bytes: SURVEY.md 8d / BASELINE.md 4).
    mult = None
    secondary = None
    extras = {}
    copy_gbps = None
    if rank == 0 and not args.no_secondary:
        # what a plain device-to-device copy of the same batch sustains on this box (read + write
        # bytes / median event time of 10): the practical ceiling SURVEY.md 8d asks the HBM
        # fractions to be quoted against as well
        dst = torch.empty_like(data)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
        for a, b in evs:
            a.record()
            dst.copy_(data)
            b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs[2:])
        copy_gbps = 2.0 * data.numel() * 8 / (ms[len(ms) // 2] * 1e-3) / 1e9
        del dst
        torch.cuda.empty_cache()
    if rank == 0 and batch == BATCH and not args.no_secondary:
        mult = timed_eltwise(hx, torch, "EltwiseMultMod(input_mod_factor=1)", N, batch, PRIMES[0])
    if rank == 0 and world == 1 and batch == BATCH and not args.no_secondary and not strong:
        del data
        torch.cuda.empty_cache()
        # Everything below is reported BESIDE the headline, which is measured and complete at
        # this point: a failure in one of these blocks is recorded in its place, never allowed to
        # cost the line.
        secondary = guarded("secondary", lambda: secondary_configs(hx, torch))

        def host_path_block():
            hp = host_path(hx)
            if not args.no_cpu_baseline:
                for key, cpu in cpu_per_call_baseline(hx).items():
                    hp[key]["cpu_baseline_one_thread"] = cpu
            return hp
        extras["host_path"] = guarded("host_path", host_path_block)
        extras["composites"] = guarded("composites", lambda: composites(hx))
        # the headline shape with the primes SEAL defaults to (60-bit)
        extras["headline_60bit"] = guarded("headline_60bit", lambda: dict(
            secondary["headline_shape_other_moduli"]["60-bit prime (Harvey60 policy)"],
            workload="in-place ForwardNTT(1,1) + InverseNTT(1,1), N=65536, batch=4096, "
                     "60-bit prime (GeneratePrimes(1, 59, false, 65536))"))
        # the C-ABI's own multi-device launcher on whatever this box shows (one std::thread +
        # stream + plans per visible GPU, one prime x 4096 polynomials each; verified against a
        # single-device run): tests/cpp/multi_device.cpp
        if os.path.exists(MULTI_DEVICE_BIN):
            torch.cuda.empty_cache()
            extras["multi_device_threads"] = guarded("multi_device_threads", lambda: run_multi_device(
                range(torch.cuda.device_count()), "weak", 5, 6, timeout=300))
    if rank == 0:
        out = {
            "metric": "Fwd+Inv NTTs/sec, N=65536 q~55b batch=4096",
            "value": value, "unit": "NTT/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "prewarm_steps": PREWARM,
            "ms_per_step": elapsed / args.steps * 1e3,
            # the same step timed with HIP events on the launch stream: median over the K
            # steps, max over ranks
            "ms_per_step_event_median": median_ms,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "per_rank_NTT_per_s": per_rank,
            # every rank's pre-timing probe (the transforms of its primes on its device against
            # the committed definition digests) and the device its plans live on (hexl_amd_ntt_device)
            "per_rank_probe_ok": per_rank_probe_ok, "per_rank_plan_device": per_rank_device,
            # what the ranks met on for the barrier and the scalar reductions (no data-path
            # collective): "nccl" (= RCCL), "gloo" (RCCL unavailable / failed its probe: why is in
            # rendezvous_note), "none" (one process)
            "launcher": "processes", "rendezvous": rv.backend or "none", "rendezvous_note": rv.note,
            "visible_devices": visible,
            "devices_shared_between_ranks": bool(oversubscribed or os.environ.get("BENCH_ONE_DEVICE") == "1"),
            "config": {
                "workload": (("BASELINE configs[3]: in-place ForwardNTT(1,1) + InverseNTT(1,1), N=65536, "
                              f"8 RNS primes (55-bit) x {batch} polynomials = {total_polys} transforms per "
                              f"direction, sharded over {world} GPU(s), resident in HBM") if strong else
                             ("in-place ForwardNTT(1,1) + InverseNTT(1,1), N=65536, "
                              f"55-bit prime per GPU, batch={batch} polys per GPU resident in HBM")),
                "N": N, "batch_per_gpu": my_polys if strong else batch, "modulus_bits": 55,
                "primes": PRIMES[:max(1, min(num_primes, 8))],
                "polynomials_total": total_polys,
                "rank0_segments": [{"prime": p, "first_poly": f, "count": c} for p, f, c in segments],
                "parallelism": (f"{args.scaling} scaling: flat (prime, polynomial) index sharded x{world}, "
                                "no collectives")},
            "hbm_algorithmic_GBps": value * 16.0 * N / 1e9,
            # the same loop held for >= 3 s (median HIP-event step of rank 0): the power-capped
            # steady state
            "sustained": sustained,
            "roofline": {
                "bound": "hbm", "kernel": dominant, "achieved": achieved,
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_source": counters_note,
                # a whole transform = two launches: algorithmic bytes of the step's transforms /
                # the event time of all its kernels / peak (per direction in `transform`); two HBM
                # round trips are inherent above N = 2^14, so the structural ceiling of this
                # figure is achievable / 2 / peak = 0.39
                "transform_frac": transform_frac, "transform": transform,
                "achievable_GBps": HBM_ACHIEVABLE_GBPS,
                "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBPS,
                "two_pass_ceiling_frac": HBM_ACHIEVABLE_GBPS / 2 / HBM_PEAK_GBPS,
                # the same fraction from the committed rocprofv3 trace of these kernel sources (its
                # average launch of the dominant kernel, a 10-step run that starts from an idle
                # GPU: a few per cent slower than the events of this run's timed region)
                "frac_by_committed_rocprof_trace": (
                    alg_bytes / ((counters.get(dominant) or {}).get("traced_avg_ms") * 1e-3) / 1e9 / HBM_PEAK_GBPS
                    if (counters.get(dominant) or {}).get("traced_avg_ms") else None),
                "copy_GBps_measured": copy_gbps,
                "frac_of_measured_copy": (achieved / copy_gbps) if copy_gbps else None,
                "algorithmic_bytes_per_launch": alg_bytes,
                "avg_kernel_ms": kern_avg, "median_kernel_ms": kern_med,
                # both rooflines per kernel: HBM (algorithmic bytes / event time, this run) and
                # instruction issue (VALU busy from the committed counters of these sources)
                "per_kernel": {k: {"ms": v, "ms_median": kern_med[k],
                                   "GBps_algorithmic": alg_bytes / (v * 1e-3) / 1e9,
                                   "frac_of_hbm_peak": alg_bytes / (v * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                                   "hbm_bytes_per_launch": (counters.get(k) or {}).get("hbm_bytes_per_launch"),
                                   "valu_busy": (counters.get(k) or {}).get("valu_busy"),
                                   "valu_insts_per_wave": (counters.get(k) or {}).get("valu_insts_per_wave"),
                                   "shader_clock_GHz": (counters.get(k) or {}).get("shader_clock_GHz"),
                                   "limiter": KERNEL_LIMITER.get(k, "hbm")}
                               for k, v in kern_avg.items()},
                "counters_source": counters_note,
                "note": ("achieved = algorithmic bytes (16*N per transform, read + write once) of the "
                         "dominant kernel's launch / its HIP-event duration on the launch stream inside "
                         "the timed region; the bound of the path is HBM, `limiter` says what holds each "
                         "kernel below it; valu_busy = SQ_ACTIVE_INST_VALU (quad-cycles) * 4 / (SIMDs * "
                         "GRBM_GUI_ACTIVE per XCD) from rocprofv3 --pmc (DESIGN.md 5)")},
        }
        if other_mode is not None:
            out[other_mode[0]] = other_mode[1]
        if mult is not None:
            out["eltwise_mult_mod"] = mult
        if secondary is not None:
            out["secondary"] = secondary
        out.update(extras)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = guarded("cpu_baseline", cpu_baseline)
        elif world > 1:
            out["cpu_baseline"] = None
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if not rv.close():
        # (an RCCL probe was left in flight on some rank: nothing more to do that is worth the
        # risk of blocking in the communicator's teardown)
        sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
