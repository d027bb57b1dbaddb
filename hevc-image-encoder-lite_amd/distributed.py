"""Multi-GPU plumbing: pictures are independent units, so ranks (one process per GPU, torch.distributed --
backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests) share no data while they encode; the only exchange
is the gather of the finished streams on one rank.  bench.py, tools/dist_batch.py and tests/test_multirank_gloo.py all go
through these functions.

    gather_streams(streams, device)     every rank contributes a list of byte strings; rank `dst` gets them all back
    gather_job_streams(job, device)     the same for the pictures of a finished Job, the slice data taken from where the kernel wrote
                                        it in HBM: no host round trip on the sending side
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


class _DeviceBytes:
    """n bytes at a device address, for torch.as_tensor (the CUDA array interface is what ROCm builds of torch implement too)"""
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}   # (torch refuses read-only views; the tensor is only read)


def gather_job_streams(job, device=None, dst=0, header=None):
    """The streams of all pictures of a finished `job` (hevc-image-encoder-lite_amd.Job), gathered on rank `dst`: a list per rank of
    `bytes`, None elsewhere.  A stream = parameter-set header + slice data; the header is a function of (qpd6, padded height, padded
    width) alone, so only a small table (length, qpd6, H, W per picture) and the slice data travel, and the slice data go from the
    job's HBM slab into the collective's send buffer by device-to-device copies (Job.device_payload) when `device` is a GPU -- the
    bytes the kernel emitted never visit this rank's host.  With device None / CPU (the gloo tests) the slice data are taken from
    job.stream(i) behind its header: same protocol, host buffers.  `header(qpd6, H, W) -> bytes` defaults to the library's."""
    import torch
    import torch.distributed as dist
    if header is None:
        from . import write_header as header
    on_gpu = device is not None and torch.device(device).type == "cuda"
    dev = torch.device(device) if on_gpu else torch.device("cpu")
    rows, parts = [], []
    for i in range(job.n):
        H, W = job.padded_size(i)
        q = int(job.qpd6[i])
        hdr = len(header(q, H, W))
        if on_gpu:
            ptr, n = job.device_payload(i)
            if ptr is None:
                raise RuntimeError(f"picture {i}: no slice data on the device (launch not finished, or host-writer mode)")
            parts.append(torch.as_tensor(_DeviceBytes(ptr, n), device=dev))
        else:
            body = job.stream(i)[hdr:]
            n = len(body)
            parts.append(torch.from_numpy(np.frombuffer(body, np.uint8).copy()))
        rows.append([n, q, H, W])
    single = not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1
    world, rank = (1, 0) if single else (dist.get_world_size(), dist.get_rank())
    total = sum(r[0] for r in rows)
    flat = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.uint8, device=dev)   # device-to-device on a GPU
    if single:
        tables, flats = [rows], [flat]
    else:
        meta = torch.tensor([len(rows), total], dtype=torch.int64, device=dev)
        metas = [torch.zeros_like(meta) for _ in range(world)]
        dist.all_gather(metas, meta)
        metas = [m.cpu().tolist() for m in metas]
        max_k, max_b = max(max(m[0] for m in metas), 1), max(max(m[1] for m in metas), 1)
        table = torch.zeros((max_k, 4), dtype=torch.int64, device=dev)
        if rows:
            table[:len(rows)] = torch.tensor(rows, dtype=torch.int64).to(dev)
        all_tables = [torch.zeros_like(table) for _ in range(world)]
        dist.all_gather(all_tables, table)
        payload = torch.zeros(max_b, dtype=torch.uint8, device=dev)
        payload[:total] = flat
        out = [torch.zeros(max_b, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
        dist.gather(payload, out, dst=dst)
        if rank != dst:
            return None
        tables = [all_tables[r][:metas[r][0]].cpu().tolist() for r in range(world)]
        flats = [out[r][:metas[r][1]] for r in range(world)]
    result = []
    for tab, fl in zip(tables, flats):
        data = fl.cpu().numpy().tobytes()
        pos, part = 0, []
        for n, q, H, W in tab:
            part.append(bytes(header(int(q), int(H), int(W))) + data[pos:pos + n])
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
