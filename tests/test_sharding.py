"""Multi-rank path on CPU (gloo, world_size 2): the (prime, poly) units are
partitioned over ranks with no data-path collective; every unit is transformed
exactly once and the gathered result equals the single-process result.  The
arithmetic is done by the oracle here (no GPU in this container); on the GPU
box the same partition drives the HIP path (bench.py --gpus N)."""
import os

import numpy as np
import pytest

from hexl_amd.sharding import max_over_ranks, shard_range, units_by_prime

N, POLYS, PRIMES_BITS = 256, 6, 30


def test_shard_range_covers_exactly_once():
    for total in (0, 1, 7, 8, 32768, 4096 * 3 + 5):
        for world in (1, 2, 3, 4, 8):
            seen = []
            for r in range(world):
                b, e = shard_range(total, world, r)
                seen += list(range(b, e))
            assert seen == list(range(total))
    # BASELINE configs[3]: 8 primes x 4096 polys over 8 GPUs -> one prime per GPU
    for g in range(8):
        b, e = shard_range(8 * 4096, 8, g)
        assert units_by_prime(b, e, 4096) == [(g, 0, 4096)]
    # 2 GPUs -> four primes each
    assert units_by_prime(*shard_range(8 * 4096, 2, 1), 4096) == [(k, 0, 4096) for k in (4, 5, 6, 7)]
    assert units_by_prime(3, 9, 4) == [(0, 3, 1), (1, 0, 4), (2, 0, 1)]


def test_strong_scaling_partition_of_config4():
    """bench.py --scaling strong: the job is always BASELINE configs[3] (8 primes x 4096
    polynomials); every (prime, polynomial) unit belongs to exactly one rank for any number of
    GPUs, whole primes for G in {1, 2, 4, 8}."""
    from hexl_amd.sharding import job_partition
    P, B = 8, 4096
    for world in (1, 2, 3, 4, 5, 8, 16):
        parts = job_partition(P, B, world, "strong")
        assert len(parts) == world
        seen = set()
        for segs in parts:
            for prime, first, count in segs:
                assert 0 <= prime < P and 0 <= first and first + count <= B and count > 0
                for u in range(prime * B + first, prime * B + first + count, 512):
                    assert u not in seen
                    seen.add(u)
                seen.add(prime * B + first + count - 1)
        assert sum(c for segs in parts for _, _, c in segs) == P * B
        sizes = [sum(c for _, _, c in segs) for segs in parts]
        assert max(sizes) - min(sizes) <= 1
        if world in (1, 2, 4, 8):  # several WHOLE primes per GPU
            for g, segs in enumerate(parts):
                per = P // world
                assert segs == [(g * per + k, 0, B) for k in range(per)]
    # weak: every rank its own prime, the job grows
    assert job_partition(8, B, 4, "weak") == [[(g, 0, B)] for g in range(4)]
    with pytest.raises(ValueError):
        job_partition(8, B, 2, "sideways")


def _worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist

    from oracle import hexl_oracle as ho
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    primes = ho.generate_primes(3, PRIMES_BITS, True, N)
    total = len(primes) * POLYS
    b, e = shard_range(total, world, rank)
    mine = []
    for prime, first, count in units_by_prime(b, e, POLYS):
        ntt = ho.NTT(N, primes[prime])
        for p in range(first, first + count):
            x = ho.fill_splitmix(N, 1000 * prime + p, primes[prime])
            mine.append(ntt.forward(x, 1, 1))
    mine = np.stack(mine) if mine else np.zeros((0, N), dtype=np.uint64)
    # timing reduction used by bench.py: the job time is the slowest rank's
    t = max_over_ranks(1.0 + rank, dist)
    assert t == float(world)
    gathered = [None] * world
    dist.all_gather_object(gathered, (b, e, mine))
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered.npy"), np.concatenate([g[2] for g in gathered]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2])
def test_two_rank_gloo_partition_matches_single_process(tmp_path, world):
    import torch.multiprocessing as mp

    from oracle import hexl_oracle as ho
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "gathered.npy"))
    primes = ho.generate_primes(3, PRIMES_BITS, True, N)
    ref = []
    for k, q in enumerate(primes):
        ntt = ho.NTT(N, q)
        for p in range(POLYS):
            ref.append(ntt.forward(ho.fill_splitmix(N, 1000 * k + p, q), 1, 1))
    assert (got == np.stack(ref)).all()


def _rendezvous_worker(rank, world, port, out_dir, prefer):
    from hexl_amd.sharding import rendezvous
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    rv = rendezvous(rank, world, local_rank=rank, prefer=prefer, probe_seconds=5.0)
    rv.barrier()
    slowest = rv.max(10.0 + rank)
    rates = rv.gather(100.0 * (rank + 1))
    with open(os.path.join(out_dir, f"rv{rank}"), "w") as f:
        f.write(repr((rv.backend, rv.note, slowest, rates)))
    rv.barrier()
    rv.close()


@pytest.mark.parametrize("prefer", ["nccl", "gloo"])
def test_rendezvous_falls_back_to_gloo_when_rccl_cannot_come_up(tmp_path, prefer):
    """bench.py's ranks meet through hexl_amd.sharding.rendezvous: RCCL when every rank brings
    it up, otherwise gloo BY ITSELF (round-3 review: an RCCL init failure must not lose the
    scaling run of a collective-free job).  Here there is no GPU, so asking for "nccl" has to
    end on gloo with the reason recorded, on every rank alike, and the barrier and the two
    reductions bench.py uses must work."""
    import ast

    import torch.multiprocessing as mp
    port = 31000 + (os.getpid() % 2000) + (7 if prefer == "gloo" else 0)
    mp.spawn(_rendezvous_worker, args=(2, port, str(tmp_path), prefer), nprocs=2, join=True)
    for rank in range(2):
        backend, note, slowest, rates = ast.literal_eval((tmp_path / f"rv{rank}").read_text())
        assert backend == "gloo"
        assert slowest == 11.0 and rates == [100.0, 200.0]
        if prefer == "nccl":
            assert note and ("GPU" in note or "NCCL" in note or "RCCL" in note), note
        else:
            assert note == "gloo requested"
    # one process: nothing to meet on
    from hexl_amd.sharding import rendezvous
    rv = rendezvous(0, 1)
    assert rv.backend is None and rv.max(3.0) == 3.0 and rv.gather(2.0) == [2.0]
    rv.barrier()
    rv.close()


def test_bench_self_launcher_builds_the_rank_environment(tmp_path):
    """`python bench.py --gpus N` outside torchrun re-runs itself under
    torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1; the ranks see
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (checked by running a stub through the very
    command the launcher builds)."""
    import subprocess
    import sys

    import bench

    cmd = bench.launcher_command(2, ["--gpus", "2", "--steps", "3"], port=29123)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29123"
    assert cmd[-5].endswith("bench.py") and cmd[-4:] == ["--gpus", "2", "--steps", "3"]
    # run a stub in place of bench.py through the same launcher line
    stub = tmp_path / "stub.py"
    stub.write_text(
        "import os, sys\n"
        "open(os.path.join(sys.argv[1], 'rank%s' % os.environ['RANK']), 'w').write(\n"
        "    ' '.join(os.environ[k] for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR')))\n")
    cmd = bench.launcher_command(2, [str(tmp_path)], port=bench.free_port())
    cmd[cmd.index(os.path.abspath(bench.__file__))] = str(stub)
    subprocess.check_call(cmd, timeout=300)
    assert (tmp_path / "rank0").read_text() == "0 0 2 127.0.0.1"
    assert (tmp_path / "rank1").read_text() == "1 1 2 127.0.0.1"


def test_bench_counter_profile_is_tied_to_the_kernel_sources(tmp_path, monkeypatch):
    """bench.py reports the counter-derived figures (HBM bytes per launch, VALU busy, shader clock)
    only when the committed profile was measured on the kernel sources of this checkout (round-2
    advisor: a roofline fraction read from a stale profile does not follow code changes)."""
    import json

    import bench
    assert bench.median([3.0, 1.0, 2.0]) == 2.0 and bench.median([4.0, 1.0, 2.0, 3.0]) == 2.5
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
    counters, note = bench.counter_profile()
    path = os.path.join(bench.ROOT, "profiles", bench.COUNTER_PROFILE)
    committed = json.load(open(path)) if os.path.exists(path) else None
    if committed is None:
        assert counters == {} and "no counter profile" in note
    elif committed["kernel_source_sha16"] == h:
        assert counters["ntt_fwd_tile_pass_bottom"]["valu_busy"] > 0.5 and "these kernel sources" in note
        for fam in ("ntt_fwd_strided_pass", "ntt_fwd_tile_pass_bottom", "ntt_inv_tile_pass_bottom",
                    "ntt_inv_strided_pass"):
            # one launch reads and writes 4096 polynomials of 512 KiB once: traffic ~ algorithmic bytes
            assert abs(counters[fam]["hbm_bytes_per_launch"] / (16.0 * 65536 * 4096) - 1.0) < 0.01
    else:
        assert counters == {} and "other kernel sources" in note
    # a profile of other sources is not reported
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    os.makedirs(tmp_path / "hexl_amd" / "csrc")
    for name in bench.KERNEL_SOURCES:
        (tmp_path / "hexl_amd" / "csrc" / name).write_text("// " + name)
    (tmp_path / "profiles" / bench.COUNTER_PROFILE).write_text(json.dumps(
        {"kernel_source_sha16": "0" * 16, "by_bench_kernel_family": {"k": {"valu_busy": 1.0}}}))
    assert bench.counter_profile()[0] == {}
    (tmp_path / "profiles" / bench.COUNTER_PROFILE).write_text(json.dumps(
        {"kernel_source_sha16": bench.kernel_source_hash(),
         "by_bench_kernel_family": {"k": {"valu_busy": 1.0}}}))
    assert bench.counter_profile()[0] == {"k": {"valu_busy": 1.0}}


def test_bench_threads_launcher_line_shape(monkeypatch, capsys):
    """`bench.py --launcher threads` prints the same JSON shape as the one-process-per-GPU launcher
    (VERDICT r3 item 1); checked here without a GPU by feeding it a result of tests/cpp/multi_device."""
    import argparse
    import json

    import bench
    fake = {"launcher": "threads", "ok": True, "error": "", "n_gpus": 2, "devices": [0, 1],
            "visible_devices": 2, "scaling": "strong", "N": 65536, "batch": 4096, "primes": 8,
            "polynomials_total": 32768, "steps": 3, "warmup": 1, "value": 4.9e6, "unit": "NTT/s",
            "ms_per_step": 13.4, "per_rank_NTT_per_s": [2.45e6, 2.46e6], "per_rank_polynomials": [16384, 16384],
            "probe_polynomials_compared": 8, "probe_mismatches": 0, "ref_device": 0}
    seen = {}

    def fake_run(devices, scaling, steps, warmup, batch=4096, n=65536, timeout=900):
        seen.update(devices=list(devices), scaling=scaling, steps=steps, batch=batch)
        return fake
    monkeypatch.setattr(bench, "run_multi_device", fake_run)
    monkeypatch.delenv("BENCH_ONE_DEVICE", raising=False)
    bench.threads_main(argparse.Namespace(gpus=2, scaling="strong", steps=3, warmup=1, batch=4096))
    line = json.loads(capsys.readouterr().out.strip().splitlines()[-1])
    assert seen == {"devices": [0, 1], "scaling": "strong", "steps": 3, "batch": 4096}
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "per_rank_NTT_per_s", "launcher",
                "rendezvous", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["launcher"] == "threads" and line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["per_rank_NTT_per_s"] == fake["per_rank_NTT_per_s"] and line["value"] == fake["value"]
    assert line["metric"].startswith("Fwd+Inv NTTs/sec") and line["dtype"] == "u64"
    assert line["verified"]["probe_mismatches"] == 0
    # a one-GPU box: the ranks share device 0
    monkeypatch.setenv("BENCH_ONE_DEVICE", "1")
    bench.threads_main(argparse.Namespace(gpus=3, scaling="weak", steps=2, warmup=1, batch=128))
    capsys.readouterr()
    assert seen["devices"] == [0, 0, 0] and seen["scaling"] == "weak" and seen["batch"] == 128
