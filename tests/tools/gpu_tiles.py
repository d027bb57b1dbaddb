#!/usr/bin/env python3
"""BASELINE.json configs[3] / SURVEY.md 8(d) config 4 on one GPU: ONE 16384x16384 G2-16 picture (seed 0x1234ABCD) as
independent tile-pictures (default 1024x1024 -> 256 tiles = one workgroup per compute unit), read in place from the
picture resident in HBM (hevce_job_set_device_window).  A sample of tiles is checked against the oracle's encode of
that window stored as its own picture -- the only definition of bit-exactness a tiled picture has (the reference crops
at 8192x8192).   usage: gpu_tiles.py [size] [tile] [tiles-to-check]"""
import os, sys, time, json, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import hevce_pkg, hevce_workload as wl
hevce = hevce_pkg.load()
size = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
ncheck = int(sys.argv[3]) if len(sys.argv) > 3 else 3
t = time.time(); img = wl.gen_g2(size, size, 16); gen_s = time.time() - t
print(f"generated {size}x{size} in {gen_s:.1f} s", file=sys.stderr, flush=True)
d_img = torch.from_numpy(img).to("cuda:0"); torch.cuda.synchronize()
grid = hevce.tile_grid(size, size, tile, tile)
with hevce.Job([(th, tw) for _, _, th, tw in grid], 2) as job:
    for i, (y, x, th, tw) in enumerate(grid):
        job.set_device_window(i, d_img.data_ptr() + y * size + x, size, keepalive=d_img)
    t = time.time(); job.run(); wall = time.time() - t
    tm = job.timings()
    streams = [job.stream(i) for i in range(len(grid))]
    picks = sorted(set([0, len(grid) // 2 + 3, len(grid) - 1][:ncheck]))
    rcons = {i: job.rcon(i) for i in picks}
import orc
checked = []
for i in picks:
    y, x, th, tw = grid[i]
    t = time.time(); s, r = orc.encode(np.ascontiguousarray(img[y:y + th, x:x + tw]), 2)
    checked.append(dict(tile=i, origin=[y, x], bytes=len(streams[i]), stream_identical=s == streams[i],
                        reconstruction_identical=bool(np.array_equal(r, rcons[i])), oracle_s=round(time.time() - t, 1)))
    print(f"tile {i}: {checked[-1]}", file=sys.stderr, flush=True)
px = size * size
line = dict(picture=f"{size}x{size} G2-16 seed 0x1234ABCD", qpd6=2, tiles=len(grid), tile=f"{tile}x{tile}",
            ctus=sum(((th + 31) // 32) * ((tw + 31) // 32) for _, _, th, tw in grid), kernel_s=round(tm["kernel_ms"] / 1e3, 2),
            fetch_s=round(tm["writer_ms"] / 1e3 + tm["d2h_ms"] / 1e3, 2), wall_s=round(wall, 2),
            mpix_per_s_kernel=round(px / tm["kernel_ms"] / 1e3, 2), mpix_per_s_wall=round(px / wall / 1e6, 2),
            total_stream_bytes=sum(len(s) for s in streams),
            sha256_of_concatenated_streams=hashlib.sha256(b"".join(streams)).hexdigest(), checked_against_oracle=checked)
print(json.dumps(line), flush=True)
