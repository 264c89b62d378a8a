"""Batch sharding of independent (prime, polynomial) transforms over ranks.

The hot path has no data dependence between polynomials, so multi-GPU is a pure
partition of the flat unit index ``u = prime * polys_per_prime + poly`` into
contiguous ranges, one per rank (the per-modulus loop of
hexl/experimental/seal/key-switch-internal.cpp:52-90 is the in-tree caller with
this shape).  No collective touches the data path; ``torch.distributed`` is used
only for the barrier and the max-over-ranks timing of the benchmark.
"""


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


def max_over_ranks(seconds, dist=None, device=None):
    """Elapsed time of the slowest rank (the whole job's time)."""
    if dist is None or not dist.is_initialized():
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, dist=None, device=None):
    """[value of rank 0, value of rank 1, ...] on every rank (bench.py: per-rank rates)."""
    if dist is None or not dist.is_initialized():
        return [float(value)]
    import torch
    mine = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
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
