"""Batch sharding across the GPUs of one node.

Images are independent units (GroupNorm/LayerNorm are per-sample, BN is in
eval mode, attention never crosses samples), so the path shards with NO
data-path collective: one process per GPU (torch.distributed over RCCL/xGMI),
each rank owns the contiguous slice [lo, hi) of the global batch and a full
weight replica (~0.9 GB fp32).  RCCL is used only for the barrier / max-time
reduction of the benchmark and for the optional gather of finished uint8
images to rank 0.
"""
import torch


def shard_range(n_items, rank, world):
    """Contiguous split; the first (n_items % world) ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(data, rank, world):
    """Slices every per-sample entry of a feed_data dict."""
    n = len(data['img_name'])
    lo, hi = shard_range(n, rank, world)
    return {k: v[lo:hi] for k, v in data.items()}


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(value, world, device):
    if world == 1:
        return value
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item()


def gather_images(u8, world, dst=0):
    """Gathers per-rank uint8 image batches [b_r,H,W,3] on rank `dst` (ragged
    shards allowed).  Returns the concatenated batch on dst, None elsewhere."""
    if world == 1:
        return u8
    import torch.distributed as dist
    rank = dist.get_rank()
    sizes = [torch.zeros(1, dtype=torch.int64, device=u8.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([u8.shape[0]], dtype=torch.int64, device=u8.device))
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = torch.zeros((mx, ) + tuple(u8.shape[1:]), dtype=u8.dtype, device=u8.device)
    pad[:u8.shape[0]] = u8
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0)


def gather_floats(value, world, device):
    """One float per rank -> list on every rank (per-rank step times, checksums)."""
    if world == 1:
        return [float(value)]
    import torch.distributed as dist
    mine = torch.tensor([value], dtype=torch.float64, device=device)
    bufs = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(bufs, mine)
    return [float(b.item()) for b in bufs]


def broadcast_state_dicts(sds, world, device, src=0):
    """Checkpoints are read / synthesised ONCE, on rank `src`, and reach the other ranks of the
    node over RCCL/xGMI ("optional broadcast of repacked weights from rank 0 at
    init") instead of N processes each re-reading or re-synthesising ~0.9 GB on the host: one flat
    buffer PER DTYPE, so integer buffers (BatchNorm counters, indices) and half-precision entries
    travel in their own type, bit for bit.  `sds`: dict module -> state_dict on `src`, ignored
    (may be None) elsewhere.  Returns the same structure with CPU tensors on every rank."""
    if world == 1:
        return sds
    import torch.distributed as dist
    rank = dist.get_rank()
    meta = [None]
    if rank == src:
        meta[0] = [(m, k, tuple(v.shape), str(v.dtype)) for m, sd in sds.items() for k, v in sd.items()]
    dist.broadcast_object_list(meta, src=src)
    meta = meta[0]
    numel = [int(torch.Size(shape).numel()) for _, _, shape, _ in meta]
    groups = {}
    for i, (_, _, _, dtype) in enumerate(meta):
        groups.setdefault(dtype, []).append(i)
    out = {}
    for dtype in sorted(groups):
        idxs = groups[dtype]
        td = getattr(torch, dtype.split('.')[-1])
        wire = torch.uint8 if td == torch.bool else td  # (no bool collectives)
        flat = torch.empty(sum(numel[i] for i in idxs), dtype=wire, device=device)
        if rank == src:
            off = 0
            for i in idxs:
                m, k = meta[i][0], meta[i][1]
                flat[off:off + numel[i]].copy_(sds[m][k].reshape(-1).to(wire))
                off += numel[i]
        dist.broadcast(flat, src=src)
        if rank == src:
            continue
        host, off = flat.cpu(), 0
        for i in idxs:
            m, k, shape, _ = meta[i]
            out.setdefault(m, {})[k] = host[off:off + numel[i]].view(shape).to(td)
            off += numel[i]
    if rank == src:
        return sds
    # module / key order of the source (state_dict order matters to strict loaders)
    return {m: {k: out[m][k] for mm, k, _, _ in meta if mm == m} for m in dict.fromkeys(mm for mm, *_ in meta)}
