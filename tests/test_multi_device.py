"""Multi-device on the C-ABI ("one host thread + one stream per GPU"): the C++
caller tests/cpp/multi_device.cpp drives every device from ONE process -- a std::thread, a
stream and its own plans per GPU, include/hexl_amd.h only -- over the shards of the flat
(prime, polynomial) index, and bit-compares every shard's first and last polynomial with a
single-device run.  Without a GPU: the binary builds, its partition is hexl_amd.sharding's, and
it fails loudly.  On the one-GPU box the device list repeats device 0 (several worker threads,
none of which ever sets a current device)."""
import ctypes as C
import json
import os
import subprocess
import threading

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "multi_device")


def run(*args, check=True):
    r = subprocess.run([BIN] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
    if check:
        assert r.returncode == 0, r.stdout + r.stderr
    return r


def test_partition_is_the_sharding_rule():
    from hexl_amd.sharding import job_partition
    assert os.path.exists(BIN), "build with python -c 'import __graft_entry__ as g; g.build()'"
    for world in (1, 2, 3, 4, 5, 8):
        devs = ",".join(str(g) for g in range(world))
        for scaling in ("strong", "weak"):
            got = json.loads(run("--print-partition", "--devices", devs, "--scaling", scaling,
                                 "--batch", 4096, "--primes", 8).stdout)
            want = job_partition(8, 4096, world, scaling)
            assert [[tuple(s) for s in segs] for segs in got] == want
    # a shard that cuts through primes
    got = json.loads(run("--print-partition", "--devices", "0,0,0", "--batch", 10, "--primes", 2).stdout)
    assert [[tuple(s) for s in segs] for segs in got] == job_partition(2, 10, 3, "strong")


def test_fails_loudly_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    r = run("--n", 4096, "--batch", 4, "--primes", 2, check=False)
    assert r.returncode == 3 and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("devices,scaling,n,batch,primes", [
    ("0", "strong", 4096, 64, 3),          # one worker, several whole primes (RNS entry point)
    ("0,0", "strong", 4096, 64, 4),        # two workers, two whole primes each
    ("0,0,0", "strong", 8192, 50, 2),      # shards that cut through primes
    ("0,0", "weak", 65536, 64, 8),         # the headline shape, a prime per worker
    ("0,0,0,0", "strong", 65536, 16, 8),   # configs[3] in small: 8 primes over 4 workers
])
def test_threads_drive_the_devices_and_match_a_single_device_run(devices, scaling, n, batch, primes):
    out = json.loads(run("--devices", devices, "--scaling", scaling, "--n", n, "--batch", batch,
                         "--primes", primes, "--steps", 2, "--warmup", 1).stdout.strip().splitlines()[-1])
    workers = len(devices.split(","))
    assert out["ok"] and out["launcher"] == "threads" and out["n_gpus"] == workers
    assert out["probe_mismatches"] == 0 and out["probe_polynomials_compared"] == 4 * workers
    assert len(out["per_rank_NTT_per_s"]) == workers and all(v > 0 for v in out["per_rank_NTT_per_s"])
    total = (workers if scaling == "weak" else primes) * batch
    assert out["polynomials_total"] == total and sum(out["per_rank_polynomials"]) == total
    assert out["value"] > 0


@pytest.mark.gpu
def test_bench_threads_launcher_emits_the_bench_line():
    """bench.py --launcher threads: the JSON shape of the one-process-per-GPU launcher."""
    import sys
    env = dict(os.environ, BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launcher", "threads",
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "128"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "per_rank_NTT_per_s", "launcher",
                "rendezvous"):
        assert key in line
    assert line["launcher"] == "threads" and line["n_gpus"] == 2 and len(line["per_rank_NTT_per_s"]) == 2
    assert line["verified"]["probe_mismatches"] == 0 and line["value"] > 0
    # round 6: every worker proved its plans on its device against the committed definition digests
    assert line["per_rank_probe_ok"] == [True, True] and line["per_rank_plan_device"] == [0, 0]


def _fixture_probes():
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "ntt_definition_fixtures.json")))["cases"]
    return [(c["q"], c["forward"]["sha256_le_u64"], c["inverse"]["sha256_le_u64"]) for c in cases if c["n"] == 65536]


@pytest.mark.gpu
def test_worker_probe_against_the_definition_digests():
    """Round 6: before it is timed every worker transforms splitmix64(1) / splitmix64(1001) mod q
    with each of ITS plans on ITS device and compares the SHA-256 with the digests of
    tests/golden/ntt_definition_fixtures.json (committed data, pinned to the big-integer definition
    of the transform: no oracle in the process).  All 8 RNS primes of configs[3] over 4 workers
    pass; a digest that belongs to another prime (what a worker that picked up the wrong prime's
    tables would produce) fails the run loudly."""
    probes = _fixture_probes()
    assert len(probes) == 8
    args = ["--devices", "0,0,0,0", "--scaling", "strong", "--n", 65536, "--batch", 8, "--primes", 8,
            "--steps", 1, "--warmup", 0]
    good = sum((["--probe", f"{q}:{f}:{i}"] for q, f, i in probes), [])
    out = json.loads(run(*args, *good).stdout.strip().splitlines()[-1])
    assert out["ok"] and out["per_rank_probe_ok"] == [True] * 4 and out["per_rank_plan_device"] == [0] * 4
    # prime 3's digests filed under prime 2: worker 1 (primes 2, 3) must fail, and with it the run
    swapped = list(probes)
    swapped[2] = (probes[2][0], probes[3][1], probes[3][2])
    bad = sum((["--probe", f"{q}:{f}:{i}"] for q, f, i in swapped), [])
    r = subprocess.run([BIN] + [str(a) for a in args + bad], capture_output=True, text=True, timeout=600)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode != 0 and not out["ok"] and "probe" in out["error"]
    assert out["per_rank_probe_ok"] == [True, False, True, True]


@pytest.mark.gpu
def test_bench_process_launcher_probe_and_wrong_device(tmp_path):
    """The one-process-per-GPU launcher (what SCALE_rNN.json measures), two ranks on one device:
    the line carries per_rank_probe_ok / per_rank_plan_device; told to expect its plans on ANOTHER
    device than the one they were created on (BENCH_PROBE_EXPECT_DEVICE: a rank that ended up on the
    wrong GPU) the run fails before anything is timed."""
    import sys
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
            "--batch", "64", "--no-cpu-baseline", "--no-secondary"]
    env = dict(os.environ, BENCH_ONE_DEVICE="1", BENCH_BACKEND="gloo", BENCH_SUSTAINED_S="0")
    r = subprocess.run(base, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([x for x in r.stdout.strip().splitlines() if x.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["per_rank_probe_ok"] == [True, True]
    assert line["per_rank_plan_device"] == [0, 0]
    r = subprocess.run(base, capture_output=True, text=True, timeout=900,
                       env=dict(env, BENCH_PROBE_EXPECT_DEVICE="1"))
    assert r.returncode != 0 and "per-rank probe FAILED" in (r.stdout + r.stderr)


@pytest.mark.gpu
def test_stream_entry_points_need_no_current_device():
    """A fresh thread (current device never set) creates a stream for device 0 and runs a plan,
    an element-wise op and a copy on it: plans carry their device, everything else runs on the
    device that owns the stream (include/hexl_amd.h, "Devices and streams")."""
    import numpy as np

    import hexl_amd as hx
    from oracle import hexl_oracle as ho
    n = 4096
    q = ho.generate_primes(1, 49, True, n)[0]
    x = ho.fill_splitmix(n, 5, q)
    result = {}

    def worker():
        lib = hx.lib
        st, plan, d = C.c_void_p(), C.c_void_p(), C.c_void_p()
        rc = lib.hexl_amd_stream_create(C.byref(st), 0)
        rc = rc or lib.hexl_amd_ntt_create(C.byref(plan), n, q, 0, 0)
        rc = rc or lib.hexl_amd_device_alloc(C.byref(d), n * 8, 0)
        rc = rc or lib.hexl_amd_copy(d, x.ctypes.data_as(C.c_void_p), n * 8, st, 0)
        rc = rc or lib.hexl_amd_ntt_forward(plan, d, d, 1, 1, 1, st)
        rc = rc or lib.hexl_amd_eltwise_mult_mod(d, d, d, n, q, 1, st)
        out = np.empty(n, dtype=np.uint64)
        rc = rc or lib.hexl_amd_copy(out.ctypes.data_as(C.c_void_p), d, n * 8, st, 1)
        dev = C.c_int(-1)
        rc = rc or lib.hexl_amd_get_device(C.byref(dev))
        lib.hexl_amd_ntt_destroy(plan)
        lib.hexl_amd_device_free(d)
        rc = rc or lib.hexl_amd_stream_destroy(st)
        result.update(rc=rc, out=out, err=lib.hexl_amd_last_error().decode(), dev=dev.value)

    t = threading.Thread(target=worker)
    t.start()
    t.join()
    assert result["rc"] == 0, result["err"]
    f = ho.NTT(n, q).forward(x, 1, 1)
    assert (result["out"] == ho.eltwise_mult_mod(f, f, q, 1)).all()
    assert result["dev"] == 0
    assert hx.lib.hexl_amd_set_device(0) == 0
    assert hx.lib.hexl_amd_stream_destroy(None) == 0
