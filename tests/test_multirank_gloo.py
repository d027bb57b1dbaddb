"""CPU: the N>1 path of bench.py's sharding/gather logic with world_size 2 over gloo (no GPU): every rank owns its own
picture (seed + rank), encodes it independently (here with the oracle standing in for the device) and rank 0 gathers the
finished streams; the gathered bytes must equal what each rank produced and what a single process produces."""
import hashlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import hevce_workload as wl
    import orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    img = wl.gen_g2(64, 96, 16, seed=0x1234ABCD + rank)          # same sharding rule as bench.py
    s, _ = orc.encode(img, 2)
    n = torch.tensor([len(s)], dtype=torch.int64)
    lens = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens, n)
    mx = int(max(int(v.item()) for v in lens))
    buf = torch.zeros(mx, dtype=torch.uint8)
    buf[:len(s)] = torch.frombuffer(bytearray(s), dtype=torch.uint8)
    out = [torch.zeros(mx, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    dist.barrier()
    if rank == 0:
        q.put([hashlib.sha256(bytes(out[r][:int(lens[r].item())].numpy())).hexdigest() for r in range(world)])
    dist.destroy_process_group()


def test_two_ranks_gather_streams():
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hevce_workload as wl
    import orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [hashlib.sha256(orc.encode(wl.gen_g2(64, 96, 16, seed=0x1234ABCD + r), 2)[0]).hexdigest() for r in range(2)]
    assert got == want and got[0] != got[1]


def _batch_worker(rank, world, port, q):
    """the sharding + collection logic of tools/dist_batch.py (LPT partition, no data-path collective, one
    all_gather_object of (index, length, sha) tables), with the oracle standing in for the device"""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import importlib
    import torch.distributed as dist
    import hevce_pkg
    import hevce_workload as wl
    import orc
    sharding = importlib.import_module(hevce_pkg.load().__name__ + ".sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = [(16 + 8 * (i % 5), 24 + 16 * (i % 3)) for i in range(11)]
    mine = sharding.partition_lpt(sizes, world)[rank]
    out = []
    for i in mine:
        s, _ = orc.encode(wl.gen_g2(sizes[i][0], sizes[i][1], 16, seed=0x1234ABCD + i), 2)
        out.append((i, len(s), hashlib.sha256(s).hexdigest()))
    table = [None] * world
    dist.all_gather_object(table, out)
    dist.barrier()
    if rank == 0:
        q.put(sorted(x for part in table for x in part))
    dist.destroy_process_group()


def test_two_ranks_share_a_batch():
    import torch.multiprocessing as mp
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hevce_workload as wl
    import orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_batch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sizes = [(16 + 8 * (i % 5), 24 + 16 * (i % 3)) for i in range(11)]
    want = []
    for i, (h, w) in enumerate(sizes):
        s, _ = orc.encode(wl.gen_g2(h, w, 16, seed=0x1234ABCD + i), 2)
        want.append((i, len(s), hashlib.sha256(s).hexdigest()))
    assert got == want
