"""Frame sharding of an All-Intra job over ranks.

Every picture of an All-Intra sequence is independent (IntraPeriod 1, BIN/encoder_intra.cfg) and so is every tile
of a picture (contexts and neighbour availability reset at tile starts, EL/EncSlice.cpp:1640-1647), so ranks get
disjoint frames and there is NO data-path collective.  The only exchanges are control-plane: a barrier around the
timed region, MAX of the elapsed time, and (optionally) gathering the per-CTU summaries on rank 0.
"""
import time


def frames_of_rank(n_frames, rank, world):
    """Contiguous block partition of POC 0..n_frames-1 (earlier ranks take the remainder)."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return list(range(lo, lo + base + (1 if rank < rem else 0)))


def _dist(world):
    if world <= 1:
        return None
    import torch.distributed as dist
    return dist


def barrier(world, device_sync=None):
    if device_sync is not None:
        device_sync()
    d = _dist(world)
    if d is not None:
        d.barrier()


def max_over_ranks(value, world, device=None):
    d = _dist(world)
    if d is None:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
    d.all_reduce(t, op=d.ReduceOp.MAX)
    return float(t.item())


def timed_steps(step, steps, warmup, world, device_sync=None, device=None):
    """bench contract: `warmup` untimed steps, then exactly `steps` steps bracketed by device sync + barrier on both
    sides; returns (MAX elapsed seconds over ranks, list of per-step return values of this rank)."""
    for _ in range(warmup):
        step()
    barrier(world, device_sync)
    t0 = time.perf_counter()
    outs = [step() for _ in range(steps)]
    barrier(world, device_sync)
    return max_over_ranks(time.perf_counter() - t0, world, device), outs


def gather_ctu_results(local, world):
    """rank 0 receives {poc: per-CTU result array} of every rank (control plane, object gather)."""
    d = _dist(world)
    if d is None:
        return dict(local)
    bucket = [None] * world if d.get_rank() == 0 else None
    d.gather_object(dict(local), bucket, dst=0)
    if bucket is None:
        return None
    merged = {}
    for part in bucket:
        merged.update(part)
    return merged
