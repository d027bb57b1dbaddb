"""CPU: the N>1 paths with world_size 2 over gloo (no GPU), through the SAME functions bench.py and tools/dist_batch.py use
(hevc-image-encoder-lite_amd/distributed.py: gather_streams, shard_batch, encode_shard) -- only the encoder is replaced by the
oracle, which stands in for the device here.  The gathered bytes must equal what a single process produces."""
import hashlib
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(rank, world, port):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import hevce_pkg
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist, importlib.import_module(hevce_pkg.load().__name__ + ".distributed")


def _worker(rank, world, port, q):
    dist, D = _setup(rank, world, port)
    import hevce_workload as wl
    import orc
    img = wl.gen_g2(64, 96, 16, seed=0x1234ABCD + rank)          # bench.py: every rank its own picture (seed + rank)
    s, _ = orc.encode(img, 2)
    got = D.gather_streams([s])                                  # bench.py's gather (RCCL there, gloo here)
    dist.barrier()
    if rank == 0:
        q.put([hashlib.sha256(part[0]).hexdigest() for part in got])
    else:
        assert got is None
    dist.destroy_process_group()


def _run(target, world=2, base_port=29500):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = base_port + (os.getpid() % 2000)
    procs = [ctx.Process(target=target, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_two_ranks_gather_streams():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hevce_workload as wl
    import orc
    got = _run(_worker)
    want = [hashlib.sha256(orc.encode(wl.gen_g2(64, 96, 16, seed=0x1234ABCD + r), 2)[0]).hexdigest() for r in range(2)]
    assert got == want and got[0] != got[1]


SIZES = [(16 + 8 * (i % 5), 24 + 16 * (i % 3)) for i in range(11)]


def _batch_worker(rank, world, port, q):
    """config 3 (a batch dealt out over the ranks): shard_batch + encode_shard + gather_streams, as tools/dist_batch.py
    and bench.py's config-3 leg do; ragged shards (11 pictures over 2 ranks, launches of 4)"""
    dist, D = _setup(rank, world, port)
    import hevce_workload as wl
    import orc
    mine = D.shard_batch(SIZES, world)[rank]
    enc = lambda part: [orc.encode(wl.gen_g2(SIZES[i][0], SIZES[i][1], 16, seed=0x1234ABCD + i), 2)[0] for i in part]
    streams = D.encode_shard(enc, mine, per_launch=4)
    got = D.gather_streams([streams[i] for i in mine])
    dist.barrier()
    if rank == 0:
        table = D.shard_batch(SIZES, world)
        q.put(sorted((i, hashlib.sha256(s).hexdigest()) for r in range(world) for i, s in zip(table[r], got[r])))
    dist.destroy_process_group()


def test_two_ranks_share_a_batch():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hevce_workload as wl
    import orc
    got = _run(_batch_worker, base_port=31500)
    want = [(i, hashlib.sha256(orc.encode(wl.gen_g2(h, w, 16, seed=0x1234ABCD + i), 2)[0]).hexdigest()) for i, (h, w) in enumerate(SIZES)]
    assert got == want


def _empty_worker(rank, world, port, q):
    """a rank without pictures (more ranks than pictures) takes part in the gather with an empty list"""
    dist, D = _setup(rank, world, port)
    got = D.gather_streams([b"abc", b""] if rank == 1 else [])
    dist.barrier()
    if rank == 0:
        q.put(got)
    dist.destroy_process_group()


def test_gather_with_an_empty_rank():
    assert _run(_empty_worker, base_port=33500) == [[], [b"abc", b""]]


class _FinishedJob:
    """what gather_job_streams reads of a finished hevc-image-encoder-lite_amd.Job, the oracle standing in for the device"""
    def __init__(self, imgs, q):
        import orc
        self.n, self.qpd6 = len(imgs), [q] * len(imgs)
        self._s = [orc.encode(im, q)[0] for im in imgs]
        self._pad = [((im.shape[0] + 31) // 32 * 32, (im.shape[1] + 31) // 32 * 32) for im in imgs]

    def padded_size(self, i):
        return self._pad[i]

    def stream(self, i):
        return self._s[i]


def _job_worker(rank, world, port, q):
    """gather_job_streams: the size table + slice data travel, the receiver writes the parameter-set headers itself (the library's
    hevce_write_header, which needs no device); rank 1 has two pictures of different sizes, rank 0 one"""
    dist, D = _setup(rank, world, port)
    import hevce_workload as wl
    imgs = [wl.gen_g2(40 + 24 * k, 64 + 32 * rank, 16, seed=0x77 + 10 * rank + k) for k in range(1 + rank)]
    job = _FinishedJob(imgs, 1 + rank)
    got = D.gather_job_streams(job)
    dist.barrier()
    if rank == 0:
        q.put([[hashlib.sha256(s).hexdigest() for s in part] for part in got])
    else:
        assert got is None
    dist.destroy_process_group()


def test_two_ranks_gather_job_streams():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hevce_workload as wl
    import orc
    got = _run(_job_worker, base_port=35500)
    want = [[hashlib.sha256(orc.encode(wl.gen_g2(40 + 24 * k, 64 + 32 * r, 16, seed=0x77 + 10 * r + k), 1 + r)[0]).hexdigest() for k in range(1 + r)] for r in range(2)]
    assert got == want
