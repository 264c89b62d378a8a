"""Batch sharding of independent (prime, polynomial) transforms over ranks.

The hot path has no data dependence between polynomials, so multi-GPU is a pure
partition of the flat unit index ``u = prime * polys_per_prime + poly`` into
contiguous ranges, one per rank (the per-modulus loop of
hexl/experimental/seal/key-switch-internal.cpp:52-90 is the in-tree caller with
this shape).  No collective touches the data path; ``torch.distributed`` is used
only for the barrier and the max-over-ranks timing of the benchmark, and that
rendezvous falls back from RCCL to gloo by itself (``rendezvous``).
"""
import os


def shard_range(total_units, world_size, rank):
    """Contiguous [begin, end) of the flat unit index owned by ``rank``."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    begin = rank * total_units // world_size
    end = (rank + 1) * total_units // world_size
    return begin, end


def units_by_prime(begin, end, polys_per_prime):
    """Split a unit range into [(prime_index, first_poly, count), ...]."""
    out = []
    u = begin
    while u < end:
        prime = u // polys_per_prime
        first = u % polys_per_prime
        count = min(end - u, polys_per_prime - first)
        out.append((prime, first, count))
        u += count
    return out


class Rendezvous:
    """What the ranks of a multi-process run meet on: a barrier and scalar reductions, never
    the data path.  `backend` is "nccl" (= RCCL on ROCm) or "gloo"; `group` the process group
    the barrier / reductions use; `device` where their scalar tensors live; `note` why RCCL
    is not in use when it was asked for."""

    def __init__(self, dist=None, backend=None, group=None, device=None, note=None, pending=False):
        self.dist, self.backend, self.group, self.device, self.note = dist, backend, group, device, note
        # an RCCL probe that timed out may have left a collective in flight: tearing the
        # process group down could then block (see close)
        self.pending = pending

    def barrier(self):
        if self.dist is None:
            return
        if self.backend == "nccl":
            self.dist.barrier(group=self.group, device_ids=[self.device.index])
        else:
            self.dist.barrier(group=self.group)

    def max(self, value):
        return max_over_ranks(value, self.dist, device=self.device, group=self.group)

    def gather(self, value):
        return gather_over_ranks(value, self.dist, device=self.device, group=self.group)

    def close(self):
        """Leaves the group.  Returns False when the caller should end the process with
        os._exit after flushing its output: an RCCL probe timed out on some rank and a
        collective may still be in flight, so destroying the NCCL communicator could block."""
        if self.dist is None or not self.dist.is_initialized():
            return True
        if self.pending:
            try:
                self.dist.barrier()  # the default (gloo) group: every rank has printed
            except Exception:  # noqa: BLE001
                pass
            return False
        self.dist.destroy_process_group()
        return True


def rendezvous(rank, world, local_rank=0, prefer="nccl", probe_seconds=120.0):
    """Process group of a one-process-per-GPU run that cannot lose the run to RCCL.

    The job has no data-path collective, so the ranks only need a barrier and
    two scalar reductions.  The DEFAULT group is always gloo over the launcher's TCP store on
    127.0.0.1 -- it needs no GPU, no IPC handles, no xGMI.  With prefer == "nccl" an RCCL group
    is created on top and probed with one all-reduce; only when EVERY rank saw it complete in
    time (agreed over gloo) does the run use it.  Any failure -- RCCL missing, communicator
    init throwing, the probe timing out -- leaves the run on gloo with the reason in `note`.
    MASTER_ADDR / MASTER_PORT come from the launcher's environment.
    """
    if world <= 1:
        return Rendezvous()
    import datetime
    import time

    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # a failed / hung RCCL probe must not take the process down with it
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world,
                            timeout=datetime.timedelta(seconds=600))
    if prefer != "nccl":
        return Rendezvous(dist, "gloo", None, "cpu", f"{prefer} requested")
    group, note, ok, started = None, None, 0, False
    # new_group below is a collective over the default group: every rank calls it or none does.
    # A rank without a GPU or without the backend says so over gloo FIRST, so that the others
    # do not wait inside new_group for a rank that will never enter it.
    can_try = 1
    if not torch.cuda.is_available():
        can_try, note = 0, "RuntimeError: no GPU visible to this rank"
    elif not dist.is_nccl_available():
        can_try, note = 0, "RuntimeError: this torch build has no NCCL/RCCL backend"
    everyone = torch.tensor([can_try], dtype=torch.int64)
    dist.all_reduce(everyone, op=dist.ReduceOp.MIN)
    if int(everyone.item()) == 0:
        if note is None:
            note = "another rank has no GPU or no NCCL/RCCL backend"
        return Rendezvous(dist, "gloo", None, "cpu", note)
    try:
        group = dist.new_group(backend="nccl",
                               timeout=datetime.timedelta(seconds=max(probe_seconds, 10.0)))
        dev = torch.device("cuda", local_rank)
        probe = torch.ones(1, dtype=torch.float64, device=dev)
        work = dist.all_reduce(probe, group=group, async_op=True)
        started = True  # enqueued: if it never completes it stays in flight
        done = torch.cuda.Event()
        deadline = time.monotonic() + probe_seconds
        # (poll instead of a blocking wait: a communicator that never comes up must not hang
        # the rank inside a synchronize)
        while not work.is_completed() and time.monotonic() < deadline:
            time.sleep(0.01)
        if not work.is_completed():
            raise RuntimeError(f"RCCL probe all-reduce not complete after {probe_seconds:.0f} s")
        done.record(torch.cuda.current_stream(dev))
        while not done.query() and time.monotonic() < deadline:
            time.sleep(0.01)
        if not done.query() or float(probe.item()) != float(world):
            raise RuntimeError("RCCL probe all-reduce returned a wrong sum or timed out")
        ok = 1
    except Exception as e:  # noqa: BLE001 -- whatever RCCL throws, the run goes on over gloo
        note = f"{type(e).__name__}: {e}".splitlines()[0][:300]
    # every rank must take the same road
    agreed = torch.tensor([ok], dtype=torch.int64)
    dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
    if int(agreed.item()) == 1:
        return Rendezvous(dist, "nccl", group, torch.device("cuda", local_rank), None)
    # did any rank get as far as issuing the probe?  Then its collective may never complete.
    began = torch.tensor([1 if started else 0], dtype=torch.int64)
    dist.all_reduce(began, op=dist.ReduceOp.MAX)
    if note is None:
        note = "another rank could not bring RCCL up"
    return Rendezvous(dist, "gloo", None, "cpu", note, pending=bool(int(began.item())))


def max_over_ranks(seconds, dist=None, device=None, group=None):
    """Elapsed time of the slowest rank (the whole job's time)."""
    if dist is None or not dist.is_initialized():
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def gather_over_ranks(value, dist=None, device=None, group=None):
    """[value of rank 0, value of rank 1, ...] on every rank (bench.py: per-rank rates)."""
    if dist is None or not dist.is_initialized():
        return [float(value)]
    import torch
    mine = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine, group=group)
    return [float(t.item()) for t in out]


def job_partition(num_primes, polys_per_prime, world_size, scaling="strong"):
    """The (prime, first polynomial, count) segments of every rank for a job of
    ``num_primes x polys_per_prime`` transforms.

    strong: the job is fixed (BASELINE configs[3]: 8 RNS primes x 4096 polynomials) and its
    flat unit index is cut into ``world_size`` contiguous shards (one prime per
    GPU at 8 GPUs, several primes per GPU below, parts of primes when world_size does not
    divide).  weak: every rank transforms ``polys_per_prime`` polynomials of its own prime
    (prime = rank mod num_primes), the job grows with the ranks.
    """
    if scaling == "weak":
        return [[(r % num_primes, 0, polys_per_prime)] for r in range(world_size)]
    if scaling != "strong":
        raise ValueError("scaling must be 'weak' or 'strong'")
    total = num_primes * polys_per_prime
    return [units_by_prime(*shard_range(total, world_size, r), polys_per_prime)
            for r in range(world_size)]
