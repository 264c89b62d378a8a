"""bench.py's main() end to end WITHOUT a GPU: torch.cuda and the hexl_amd compute entry points
are replaced by stand-ins that do nothing and report fixed kernel times, so that what is checked
is bench.py's own control flow and the shape of the ONE JSON line the driver parses -- every key
of the contract, the `roofline` and `cpu_baseline` objects, the blocks reported beside the
headline -- and that a failure in one of those blocks cannot cost the line.  (Numbers are
meaningless here; the GPU runs are what measure.)"""
import contextlib
import ctypes as C
import json
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

KERNEL_MS = {"ntt_fwd_strided_pass": 0.68, "ntt_fwd_tile_pass_bottom": 0.93,
             "ntt_inv_tile_pass_bottom": 0.98, "ntt_inv_strided_pass": 0.68}


class FakeTensor:
    def __init__(self, shape):
        self.shape = tuple(shape) if isinstance(shape, (tuple, list)) else (int(shape),)

    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n

    def __getitem__(self, key):
        if isinstance(key, slice):
            start, stop, _ = key.indices(self.shape[0])
            return FakeTensor((max(0, stop - start),) + self.shape[1:])
        return FakeTensor(self.shape[1:] or (1,))

    def clone(self):
        return FakeTensor(self.shape)

    def copy_(self, other):
        return self

    def repeat(self, k):
        return FakeTensor((self.shape[0] * k,) + self.shape[1:])

    def view(self, *a):
        return self

    def data_ptr(self):
        return 0

    payload = None  # (the per-rank probe's one polynomial: real words, see fake_hexl_amd)


def fake_torch():
    t = types.ModuleType("torch")
    t.int64 = "int64"
    t.empty = lambda shape, dtype=None, device=None: FakeTensor(shape)
    t.empty_like = lambda x: FakeTensor(x.shape)
    t.equal = lambda a, b: True
    t.device = lambda kind, index=0: types.SimpleNamespace(type=kind, index=index)

    class Event:
        def __init__(self, enable_timing=False):
            pass

        def record(self, stream=None):
            pass

        def elapsed_time(self, other):
            return 3.3

        def query(self):
            return True

    class Stream:
        cuda_stream = 0

        def wait_stream(self, other):
            pass

        def synchronize(self):
            pass

    class CUDAGraph:
        def replay(self):
            pass

    cuda = types.SimpleNamespace(
        Event=Event, Stream=Stream, CUDAGraph=CUDAGraph, is_available=lambda: True,
        set_device=lambda i: None, device_count=lambda: 1, synchronize=lambda: None,
        empty_cache=lambda: None, current_stream=lambda dev=None: Stream(),
        stream=lambda s: contextlib.nullcontext(), graph=lambda g, stream=None: contextlib.nullcontext())
    t.cuda = cuda
    return t


def fake_hexl_amd(real, fail_composites=False):
    """The real package's host-side number theory; every compute entry point a no-op that feeds
    the launch profiler with fixed kernel times."""
    from oracle import hexl_oracle as ho  # (tests may use the checker: the stand-in's probe transform)
    hx = types.ModuleType("hexl_amd")
    state = {"profiling": False, "records": []}

    def launched(names):
        if state["profiling"]:
            state["records"] += [(k, KERNEL_MS[k]) for k in names]

    class NTT:
        def __init__(self, n, q, root=0, device=None):
            self.n, self.q, self._h = n, q, C.c_void_p(1)

        def GetDevice(self):
            return 0

        def ComputeForward(self, out, x, a, b):
            if x.payload is not None:  # the per-rank probe: the real transform, by the oracle
                out.payload = ho.NTT(self.n, self.q).forward(x.payload, 1, 1)
            launched(["ntt_fwd_strided_pass", "ntt_fwd_tile_pass_bottom"] if self.n > 16384
                     else ["ntt_fwd_tile_pass_bottom"])

        def ComputeInverse(self, out, x, a, b):
            if x.payload is not None:
                out.payload = ho.NTT(self.n, self.q).inverse(x.payload, 1, 1)
            launched(["ntt_inv_tile_pass_bottom", "ntt_inv_strided_pass"] if self.n > 16384
                     else ["ntt_inv_tile_pass_bottom"])

    def profile_start(n=4096):
        state["profiling"], state["records"] = True, []

    def profile_stop():
        state["profiling"] = False
        return list(state["records"])

    def composites_op(*a, **k):
        if fail_composites:
            raise RuntimeError("simulated failure in a block beside the headline")

    hx.NTT = NTT
    hx.profile_start, hx.profile_stop = profile_start, profile_stop
    def fill_splitmix(data, n, batch, seed0, bound):
        # (one polynomial = the per-rank probe of bench.py: the stand-in carries its real words, so
        # that the probe's comparison with the committed definition digests is exercised here too)
        if batch == 1 and data.shape == (1, n):
            data.payload = ho.fill_splitmix(n, seed0, bound)
    hx.fill_splitmix = fill_splitmix
    hx.to_numpy = lambda t: t.payload
    for name in ("EltwiseMultMod", "EltwiseFMAMod", "EltwiseReduceMod", "EltwiseReduceFMAMod",
                 "ComputeForwardRNS", "ComputeInverseRNS"):
        setattr(hx, name, lambda *a, **k: None)
    for name in ("DyadicMultiply", "DyadicMultiplyBatch", "KeySwitch", "KeySwitchBatch"):
        setattr(hx, name, composites_op)
    hx.GeneratePrimes = real.GeneratePrimes
    counters = {"ks_graph_replays": 0}

    def get_counter(key):
        counters[key] = counters.get(key, 0) + 17  # (every look: 17 more)
        return counters[key]
    hx.get_counter = get_counter
    hx.set_tuning = lambda key, value: None
    hx.from_numpy = lambda a, device="cuda": FakeTensor(a.shape)
    hx.lib = types.SimpleNamespace(
        hexl_amd_ntt_forward_host=lambda *a: 0, hexl_amd_host_alloc=lambda *a: 1,  # (no mapped memory here)
        hexl_amd_host_free=lambda *a: 0, hexl_amd_release_stream_workspaces=lambda *a: 0)
    return hx


def run_main(monkeypatch, capfd, argv, fail_composites=False):
    import hexl_amd as real
    import hexl_amd.sharding as real_sharding

    import bench
    fake = fake_hexl_amd(real, fail_composites)
    fake.__path__ = real.__path__  # (a package: bench.py imports hexl_amd.sharding, which is torch-free)
    fake.sharding = real_sharding
    monkeypatch.setitem(sys.modules, "torch", fake_torch())
    monkeypatch.setitem(sys.modules, "hexl_amd", fake)
    monkeypatch.setitem(sys.modules, "hexl_amd.sharding", real_sharding)
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    monkeypatch.setattr(bench, "PREWARM", 1)
    monkeypatch.setattr(bench, "MULTI_DEVICE_BIN", "/nonexistent")  # (needs a GPU)
    monkeypatch.setattr(bench, "HOST_CALL_BUDGET_BIN", "/nonexistent")  # (needs a GPU)
    monkeypatch.setenv("BENCH_SUSTAINED_S", "0.05")
    # the CPU legs are real but bounded: a fraction of a second each
    monkeypatch.setattr(bench, "cpu_baseline", lambda: bench.__dict__["_cpu_baseline_real"](0.2, 0.2))
    for key in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(key, raising=False)
    saved = os.dup(1)  # main() points fd 1 at stderr for the libraries under it
    try:
        bench.main()
    finally:
        os.dup2(saved, 1)
        os.close(saved)
    out = capfd.readouterr().out.strip().splitlines()
    assert len(out) == 1, out  # exactly ONE line on stdout
    return json.loads(out[0])


@pytest.fixture(autouse=True)
def _keep_real_cpu_baseline():
    import bench
    bench.__dict__.setdefault("_cpu_baseline_real", bench.cpu_baseline)
    yield


def test_default_line_has_the_contract_and_the_round_4_blocks(monkeypatch, capfd):
    line = run_main(monkeypatch, capfd, ["--steps", "3", "--warmup", "1"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline",
                "per_rank_NTT_per_s", "launcher", "rendezvous", "sustained", "host_path", "composites",
                "headline_60bit", "secondary", "eltwise_mult_mod"):
        assert key in line, key
    assert line["metric"].startswith("Fwd+Inv NTTs/sec") and line["unit"] == "NTT/s"
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1
    assert line["higher_is_better"] is True and line["vs_baseline"] is None and line["dtype"] == "u64"
    assert "workload" in line["config"] and "model" not in line["config"]
    assert line["per_rank_probe_ok"] == [True] and line["per_rank_plan_device"] == [0]  # round 6
    roof = line["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "transform_frac", "transform",
                "achievable_GBps", "per_kernel"):
        assert key in roof, key
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and roof["kernel"] == "ntt_inv_tile_pass_bottom"
    # 4096 polynomials x 1 MiB per launch over the stand-in's 0.98 ms
    assert abs(roof["achieved"] - 16.0 * 65536 * 4096 / 0.98e-3 / 1e9) < 1.0
    assert abs(roof["frac"] - roof["achieved"] / 8000.0) < 1e-9
    assert set(roof["transform"]) == {"fwd", "inv"} and 0 < roof["transform_frac"] < roof["frac"]
    cpu = line["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["unit"] == "NTT/s" and cpu["cores"] >= 1 and cpu["value"] > 0
    assert line["sustained"]["steps"] >= 64 and line["sustained"]["ms_per_step_median"] == 3.3
    assert set(k for k in line["host_path"] if k.startswith("N=")) == {"N=4096", "N=16384", "N=65536"}
    assert line["host_path"]["N=4096"]["cpu_baseline_one_thread"]["us_per_call"] > 0
    assert "ctypes" in line["host_path"]["measured_through"]
    ks = line["composites"]["key_switch"]
    for key in ("one_target_per_call_us", "one_target_per_call_eager_us", "one_target_per_call_path",
                "ntt_floor_us", "hbm_floor_us", "one_target_frac_of_ntt_floor",
                "256_targets_frac_of_ntt_floor", "256_targets_frac_of_hbm_peak", "transforms_per_target"):
        assert key in ks, key
    # n = 16384, D = 7, C = 2: 7 + 2 inverse and 49 + 14 forward transforms per target
    assert ks["transforms_per_target"]["forward"] == 63 and ks["transforms_per_target"]["inverse"] == 9
    assert "graph_of_32" in line["secondary"]["config2"]
    assert set(line["secondary"]["headline_shape_other_moduli"]) >= {
        "57-bit prime (Lazy32 policy)", "59-bit prime (Lazy16 policy)", "60-bit prime (Harvey60 policy)"}
    assert line["headline_60bit"]["q"] == line["secondary"]["headline_shape_other_moduli"][
        "60-bit prime (Harvey60 policy)"]["q"]


def test_a_failing_side_block_does_not_cost_the_line(monkeypatch, capfd):
    line = run_main(monkeypatch, capfd, ["--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                    fail_composites=True)
    assert "simulated failure" in line["composites"]["error"]
    assert line["value"] > 0 and "roofline" in line and "secondary" in line
    assert "cpu_baseline" not in line or line["cpu_baseline"] is None
