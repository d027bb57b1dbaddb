#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: Mpixels/s encoded (bit-exact .h265) at qpd6=2.

A step = one pass of the hot path over one batch of synthetic input: every rank encodes ONE 3840x2160 G2-16 picture
(BASELINE.json configs[1]; pads to 3840x2176 = 8160 CTUs) that is already resident in HBM: analysis
analysis kernel (which also emits the slice data) -> D2H of the decision records and the slice data behind a
host-written header -> complete .h265 bytes.  Images are independent, so with
N ranks each rank encodes its own picture (seed + rank, weak scaling, no data-path collective); the only exchange is
the gather of the finished streams to rank 0 over RCCL, inside the timed region.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
ALGO_BYTES_PER_PIXEL = 4.125   # 1 B luma read + 1 B recon written + 2.125 B decision record


def cpu_baseline(img_full, qpd6):
    """The reference itself (oracle/_ref, built in the build container and shipped as a .so) timed on ONE host core
    on a bounded sample of the same workload: the top-left 768x512 crop (384 CTUs)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    sample = np.ascontiguousarray(img_full[:512, :768])
    kind = "reference" if orc.ref() is not None else "port"
    fn = orc.encode_ref if kind == "reference" else orc.encode
    t = time.perf_counter()
    stream, rcon = fn(sample, qpd6)
    dt = time.perf_counter() - t
    return {"value": round(rcon.size / dt / 1e6, 5), "unit": "Mpixels/s", "cores": 1, "kind": kind,
            "sample": f"top-left 768x512 crop (384 CTUs) of the workload image, {dt:.1f} s, {len(stream)} bytes",
            "host_cores": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--qpd6", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import hevce_pkg
    import hevce_workload as wl
    hevce = hevce_pkg.load()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    img = wl.gen_g2(args.height, args.width, 16, seed=0x1234ABCD + rank)
    d_img = torch.from_numpy(img).to(dev)                      # the input is resident in HBM before timing starts
    H, W = hevce.pad32(args.height), hevce.pad32(args.width)
    nctu = (H // 32) * (W // 32)
    stream_obj = torch.cuda.current_stream(dev)

    job = hevce.Job([img.shape], args.qpd6, device=local)
    job.set_device_image(0, d_img.data_ptr(), keepalive=d_img)
    job.set_stream(stream_obj.cuda_stream)

    gathered = {}

    def step():
        job.run()                                              # kernel + D2H + header: .h265 complete
        s = job.stream(0)
        if world > 1:                                          # gather the finished streams on rank 0 (RCCL over xGMI)
            n = torch.tensor([len(s)], dtype=torch.int64, device=dev)
            lens = [torch.zeros_like(n) for _ in range(world)]
            dist.all_gather(lens, n)
            mx = int(max(int(v.item()) for v in lens))
            buf = torch.zeros(mx, dtype=torch.uint8, device=dev)
            buf[:len(s)] = torch.frombuffer(bytearray(s), dtype=torch.uint8).to(dev)
            out = [torch.zeros(mx, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == 0 else None
            dist.gather(buf, out, dst=0)
            if rank == 0:
                for r in range(world):
                    gathered[r] = bytes(out[r][:int(lens[r].item())].cpu().numpy())
        return s

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    first = None
    for _ in range(args.warmup):
        first = step()
    sync()
    kernel_ms, writer_ms, d2h_ms = [], [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        s = step()
        t = job.timings()
        kernel_ms.append(t["kernel_ms"]); writer_ms.append(t["writer_ms"]); d2h_ms.append(t["d2h_ms"])
        if first is None:
            first = s
        assert s == first, "non-deterministic stream"
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        e = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(e, op=dist.ReduceOp.MAX)
        elapsed = float(e.item())

    pixels_per_step = H * W * world
    value = pixels_per_step * args.steps / elapsed / 1e6
    k_avg = float(np.mean(kernel_ms))
    achieved = ALGO_BYTES_PER_PIXEL * H * W / (k_avg * 1e-3) / 1e9
    # HBM traffic cannot be counted from inside this process: it comes from the committed rocprofv3 --pmc passes of this
    # kernel (profiles/pmc_traffic.json, bytes per CTU) scaled to this launch's CTU count; null when that file is absent
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            traffic = round(json.load(f)["hbm_bytes_per_ctu"] * nctu)
    except Exception:
        traffic = None
    # the bound that actually holds: VALU issue of the one compute unit a picture occupies -- from the committed
    # instruction-mix PMC passes of this kernel (profiles/r01_pmc_instmix.json); informational, next to the mandated HBM figures
    valu = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_instmix.json")) as f:
            m = json.load(f)
            valu = {"valu_issue_utilisation_of_the_cu": round(m["valu_issue_utilisation_of_the_cu"], 3),
                    "valu_insts_per_ctu": round(m["valu_insts_per_ctu"]), "cu_cycles_per_ctu": round(m["cu_cycles_per_ctu"])}
    except Exception:
        valu = None

    if rank == 0:
        line = {
            "metric": "Mpixels/s encoded (bit-exact .h265) at qpd6=2",
            "value": round(value, 4), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "i32", "data": "synthetic",
            "config": {"workload": f"one {args.width}x{args.height} G2-16 grayscale picture per GPU (seed 0x1234ABCD+rank), "
                                   f"qpd6={args.qpd6}, padded {W}x{H} = {nctu} CTUs, input resident in HBM; "
                                   "kernel (analysis + slice-data emission) + D2H + header" + ("; RCCL gather of streams to rank 0" if world > 1 else ""),
                       "images_per_gpu": 1, "parallelism": f"image-per-gpu x{world}"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 6), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 9), "traffic": traffic,
                         "kernel": "hevce_analyse_kernel", "kernel_ms_avg": round(k_avg, 3),
                         "us_per_ctu": round(k_avg * 1e3 / nctu, 2),
                         "note": "serial-dependency bound by construction (SURVEY 0.2): one picture = one workgroup chain",
                         "valu_issue": valu},
            "phases_ms": {"kernel": round(k_avg, 2), "d2h": round(float(np.mean(d2h_ms)), 2), "host_assembly": round(float(np.mean(writer_ms)), 2)},
            "stream_bytes": len(first), "stream_sha256": hashlib.sha256(first).hexdigest(),
        }
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(img, args.qpd6)
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu_1core"] = round(value / cb["value"], 2)
        if world > 1:
            line["gathered_streams"] = len(gathered)
        print(json.dumps(line), flush=True)
    job.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
