"""Multi-GPU sharding of a sample batch (host logic only, device agnostic).

Samples are independent (every pipeline register of the reference core is
per sample; the NCO phase is the closed form phase0 + n*fcw), so a batch of
n_total samples is split into contiguous blocks by GLOBAL sample index and each
rank generates / processes its own block: no scatter and no collective on the
data path.  The only collectives are after the fact:

  * reduce_digest(): all-reduce (sum mod 2^64) of per-shard digests -- digests
    are position-aware and additive, so the sum equals the digest of the whole;
  * gather_to_root(): collect the output shards on one rank (RCCL gather over
    xGMI on GPUs, gloo on CPU) when a single consumer needs them;
  * pipelined_gather(): the same, chunk by chunk, each chunk's gather enqueued
    as soon as that chunk has been computed, so that the transfer of chunk k
    overlaps the computation of chunk k+1.

The reference has no counterpart (single process); this
is new work for the 8-GPU configuration of BASELINE.json.
"""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """[start, start+count) of `rank`: contiguous, sizes differ by at most 1,
    the first n_total % world ranks get the extra sample."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, rem = divmod(n_total, world)
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def reduce_digest(local_digest, device="cpu", group=None):
    """Sum of the shards' 64-bit digests modulo 2^64 on every rank."""
    v = local_digest & 0xFFFFFFFFFFFFFFFF
    if v >= 1 << 63:
        v -= 1 << 64
    t = torch.tensor([v], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)  # wraps mod 2^64
    return int(t.item()) & 0xFFFFFFFFFFFFFFFF


def gather_to_root(shard, n_total, dst=0, group=None):
    """Collect equally typed 1-D shards (sizes per shard_range) on `dst`;
    returns the concatenation there and None elsewhere."""
    if not (dist.is_available() and dist.is_initialized()):
        return shard
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [shard_range(n_total, r, world)[1] for r in range(world)]
    width = max(counts)
    # gather needs equal sizes: pad to the widest shard
    padded = shard
    if shard.numel() < width:
        padded = torch.cat([shard, shard.new_zeros(width - shard.numel())])
    outs = None
    if rank == dst:
        outs = [torch.empty(width, dtype=shard.dtype, device=shard.device)
                for _ in range(world)]
    dist.gather(padded, outs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([o[:c] for o, c in zip(outs, counts)])


def chunk_ranges(n, chunks):
    """`chunks` contiguous [start, stop) pieces of range(n), sizes differing by
    at most one (the first n % chunks pieces are the longer ones); empty
    pieces are dropped."""
    out = []
    for k in range(chunks):
        a, c = shard_range(n, k, chunks)
        if c:
            out.append((a, a + c))
    return out


def pipelined_gather(compute_chunk, shards, chunks=8, dst=0, group=None,
                     comm_stream=None):
    """Compute and collect equally sized 1-D output shards chunk by chunk.

    compute_chunk(start, stop) enqueues the computation of samples
    [start, stop) of every tensor in `shards` (same length n on every rank) on
    the CURRENT stream.  After each chunk its slices are gathered on rank
    `dst` asynchronously -- on `comm_stream` when given (a CUDA/HIP side
    stream that first waits for the chunk's kernels), so the transfer runs
    while the next chunk is being computed.  Returns, on `dst`, one list per
    shard tensor holding every rank's full-length copy (rank order), and None
    elsewhere.  Every rank must call it with the same n and chunking.
    """
    n = shards[0].numel()
    pieces = chunk_ranges(n, chunks)
    live = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if live else 1
    rank = dist.get_rank(group) if live else 0
    gathered = None
    if rank == dst:
        gathered = [[torch.empty_like(t) for _ in range(world)] for t in shards]
    works = []
    for a, b in pieces:
        compute_chunk(a, b)
        if not live:
            if gathered is not None:
                for t, g in zip(shards, gathered):
                    g[0][a:b].copy_(t[a:b])
            continue
        ctx = None
        if comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record()                       # chunk's kernels, current stream
            comm_stream.wait_event(ev)
            ctx = torch.cuda.stream(comm_stream)
            ctx.__enter__()
        try:
            for i, t in enumerate(shards):
                outs = None
                if rank == dst:
                    outs = [g[a:b] for g in gathered[i]]
                works.append(dist.gather(t[a:b], outs, dst=dst, group=group,
                                         async_op=True))
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)
    for w in works:
        w.wait()
    return gathered
