"""Frame and tile sharding of an All-Intra job over ranks.

Every picture of an All-Intra sequence is independent (IntraPeriod 1, BIN/encoder_intra.cfg) and so is every tile
of a picture (contexts and neighbour availability reset at tile starts, EL/EncSlice.cpp:1640-1647), so ranks get
disjoint frames - or, for a job with fewer frames than ranks, disjoint tile ranges of a frame (units_of_rank) - and there is NO collective inside the search.  The exchanges are: a barrier around the timed region, MAX
of the elapsed time, (optionally) the per-CTU summaries on rank 0, and the one data-path step the job has - the final
gather of every rank's slice_data bytes on rank 0 (gather_payloads; RCCL when the process group is "nccl").
"""
import time


def frames_of_rank(n_frames, rank, world):
    """Contiguous block partition of POC 0..n_frames-1 (earlier ranks take the remainder)."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return list(range(lo, lo + base + (1 if rank < rem else 0)))


def tile_bounds(n_ctus, n_tiles):
    """first CTU column / row of every tile of a uniform grid, plus the end (TileUniformSpacing: boundary i at i * n_ctus / n_tiles, EL/EncSlice / CL/Slice.cpp
    uniform spacing rule that the handle's tile map follows)"""
    return [(i * n_ctus) // n_tiles for i in range(n_tiles + 1)]


def tile_ctus(ctus_w, ctus_h, tile_cols, tile_rows, tile):
    """CTU raster addresses of one tile of the uniform tile grid, in the tile's own raster order = the order its stream codes them in"""
    cb, rb = tile_bounds(ctus_w, tile_cols), tile_bounds(ctus_h, tile_rows)
    tx, ty = tile % tile_cols, tile // tile_cols
    return [y * ctus_w + x for y in range(rb[ty], rb[ty + 1]) for x in range(cb[tx], cb[tx + 1])]


def units_of_rank(n_frames, tiles_per_frame, rank, world):
    """Tile-level sharding for jobs with fewer frames than ranks (north_star: "independent CTUs shard across the GPUs"; a tile is the independent CTU stream).  The
    job's (frame, tile) streams in frame-major raster order are cut into `world` contiguous blocks; returns this rank's block as a list of
    (frame, first_tile, n_tiles) - at most one partial frame at either end, whole frames in between.  With tiles_per_frame == 1 this is frames_of_rank."""
    total = n_frames * tiles_per_frame
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    units = []
    while lo < hi:
        f, t = divmod(lo, tiles_per_frame)
        n = min(tiles_per_frame - t, hi - lo)
        units.append((f, t, n))
        lo += n
    return units


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


def _gather_var(t, world, device):
    """variable-length 1-D tensors of every rank on rank 0: all_gather of the lengths, then one gather of the padded tensors"""
    import torch
    d = _dist(world)
    n = torch.tensor([t.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    d.all_gather(sizes, n)
    sizes = [int(x.item()) for x in sizes]
    cap = max(1, max(sizes))
    pad = torch.zeros(cap, dtype=t.dtype, device=device)
    pad[:t.numel()] = t.to(device)
    bucket = [torch.zeros(cap, dtype=t.dtype, device=device) for _ in range(world)] if d.get_rank() == 0 else None
    d.gather(pad, bucket, dst=0)
    if bucket is None:
        return None
    return [b[:k].cpu() for b, k in zip(bucket, sizes)]


def gather_payloads(local, world, device="cpu"):
    """Final bitstream gather: local = {poc: [bytes of tile 0, tile 1, ...]} (numpy uint8 arrays) of this rank's frames; rank 0
    gets {poc: [tile payloads]} of the whole job, other ranks None.  Two tensor collectives (an index of int64 triples and the
    bytes) over the job's process group - RCCL over xGMI for backend "nccl" with device="cuda", gloo in the CPU tests."""
    import numpy as np
    import torch
    if _dist(world) is None:
        return {poc: [np.asarray(b, np.uint8) for b in tiles] for poc, tiles in local.items()}
    index, chunks = [], []
    for poc in sorted(local):
        for t, b in enumerate(local[poc]):
            index += [poc, t, len(b)]
            chunks.append(np.asarray(b, np.uint8))
    blob = np.concatenate(chunks) if chunks else np.zeros(0, np.uint8)
    idx = _gather_var(torch.tensor(index, dtype=torch.int64), world, device)
    data = _gather_var(torch.from_numpy(blob), world, device)
    if idx is None:
        return None
    merged = {}
    for ix, dat in zip(idx, data):
        ix = ix.numpy().reshape(-1, 3); dat = dat.numpy(); off = 0
        for poc, t, n in ix:
            merged.setdefault(int(poc), []).append(dat[off:off + int(n)].copy()); off += int(n)
    return merged
