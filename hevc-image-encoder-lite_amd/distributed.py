"""Multi-GPU plumbing: pictures are independent units, so ranks (one process per GPU, torch.distributed --
backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) share no data while they encode; the only exchange
is the gather of the finished streams on one rank.  bench.py, tools/dist_batch.py and tests/test_multirank_gloo.py all go
through these functions.

    gather_streams(streams, device)     every rank contributes a list of byte strings; rank `dst` gets them all back
    shard_batch(sizes, world)           which rank encodes which picture (longest-processing-time-first, sharding.py)
    encode_shard(encode, sizes, ...)    a rank's shard through a batch encoder, in launches of at most `per_launch` pictures
"""
import numpy as np

from . import sharding


def gather_streams(streams, device=None, dst=0):
    """streams: this rank's list of `bytes`.  Returns on rank `dst` a list (one entry per rank) of lists of bytes, elsewhere
    None.  Three collectives: all_gather of (count, total bytes), all_gather of the padded length tables, gather of the padded
    payloads -- tensors live on `device` (the rank's GPU for RCCL, the CPU for gloo)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [list(streams)]
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device("cpu") if device is None else device
    total = sum(len(s) for s in streams)
    meta = torch.tensor([len(streams), total], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    metas = [m.cpu().tolist() for m in metas]
    max_k, max_b = max(max(m[0] for m in metas), 1), max(max(m[1] for m in metas), 1)
    lens = torch.zeros(max_k, dtype=torch.int64, device=dev)
    if streams:
        lens[:len(streams)] = torch.tensor([len(s) for s in streams], dtype=torch.int64).to(dev)
    all_lens = [torch.zeros_like(lens) for _ in range(world)]
    dist.all_gather(all_lens, lens)
    payload = torch.zeros(max_b, dtype=torch.uint8, device=dev)
    if total:
        payload[:total] = torch.from_numpy(np.frombuffer(b"".join(streams), np.uint8).copy()).to(dev)
    out = [torch.zeros(max_b, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(payload, out, dst=dst)
    if rank != dst:
        return None
    result = []
    for r in range(world):
        k = metas[r][0]
        ls = all_lens[r][:k].cpu().tolist()
        flat = out[r][:metas[r][1]].cpu().numpy().tobytes()
        pos, part = 0, []
        for n in ls:
            part.append(flat[pos:pos + n])
            pos += n
        result.append(part)
    return result


def shard_batch(sizes, world):
    """per rank the indices of the pictures it encodes; every rank computes the same table, nothing is exchanged"""
    return sharding.partition_lpt(sizes, world)


def encode_shard(encode_batch, indices, per_launch=256):
    """encode_batch(list of picture indices) -> list of streams (one launch: one workgroup per picture, at most one per compute
    unit side by side).  Returns {index: stream} for this rank's shard."""
    out = {}
    for base in range(0, len(indices), per_launch):
        part = indices[base:base + per_launch]
        for i, s in zip(part, encode_batch(part)):
            out[i] = s
    return out
