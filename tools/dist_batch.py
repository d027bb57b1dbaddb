#!/usr/bin/env python3
"""BASELINE.json configs[2] on N GPUs of one node: a batch of independent pictures, dealt out over
the ranks by CTU count (hevc-image-encoder-lite_amd/sharding.py), every rank encoding its shard in launches of at most 256
pictures, no collective on the data path; rank 0 collects (index, length, SHA-256) of every stream
(hevc-image-encoder-lite_amd/distributed.py: shard_batch, encode_shard, gather_streams -- the functions bench.py uses).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/dist_batch.py [n] [h] [w]
(N = 1: plain `python tools/dist_batch.py ...`).  Synthetic G2-16 pictures, seed 0x1234ABCD + index; --unique K generates only
K different pictures (index mod K) to keep the host side short."""
import hashlib, importlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = int(args[0]) if len(args) > 0 else 1024
    h = int(args[1]) if len(args) > 1 else 1080
    w = int(args[2]) if len(args) > 2 else 1920
    unique = int(sys.argv[sys.argv.index("--unique") + 1]) if "--unique" in sys.argv else 8
    world, rank, local = (int(os.environ.get(k, d)) for k, d in (("WORLD_SIZE", "1"), ("RANK", "0"), ("LOCAL_RANK", "0")))
    import torch
    import torch.distributed as dist
    import hevce_pkg, hevce_workload as wl
    hevce = hevce_pkg.load()
    D = importlib.import_module(hevce.__name__ + ".distributed")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        torch.cuda.set_device(local)
    sizes = [(h, w)] * n
    mine = D.shard_batch(sizes, world)[rank]
    pics = {k: wl.gen_g2(h, w, 16, seed=0x1234ABCD + k) for k in sorted({i % unique for i in mine})}
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()

    def enc(part):                                              # one launch: one workgroup per picture (clusters when few)
        with hevce.Job([sizes[i] for i in part], 2, device=local) as job:
            for j, i in enumerate(part):
                job.upload(j, pics[i % unique])
            job.run()
            return [job.stream(j) for j in range(len(part))]
    streams = D.encode_shard(enc, mine, per_launch=256)
    out = [(i, len(streams[i]), hashlib.sha256(streams[i]).hexdigest()) for i in mine]
    elapsed = time.perf_counter() - t0
    if world > 1:
        e = torch.tensor([elapsed], dtype=torch.float64, device=torch.device("cuda", local))
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        elapsed = float(e.item())
        got = D.gather_streams([json.dumps(out).encode()], device=torch.device("cuda", local))
        out = [tuple(x) for part in got for x in json.loads(part[0].decode())] if rank == 0 else out
    if rank == 0:
        out.sort()
        ok = [i for i, _, _ in out] == list(range(n)) and all(s == out[i % unique][2] for i, _, s in out)
        H, W = hevce.pad32(h), hevce.pad32(w)
        print(json.dumps({"pictures": n, "size": f"{w}x{h}", "n_gpus": world, "wall_s": round(elapsed, 2),
                          "mpix_per_s": round(H * W * n / elapsed / 1e6, 2), "bytes": sum(l for _, l, _ in out),
                          "all_streams_collected_and_consistent": ok}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
