#!/usr/bin/env python3
"""Turn one measurement pass (tools/r02_profile_run.sh on the GPU box -> gpurun_out/r02p/) into the tracked summaries under
profiles/: rocprofv3 writes rocpd databases (sqlite) on this image; the kernel statistics and the PMC values are read from
their `top_kernels` / `kernels` / `counters_collection` views."""
import csv, glob, json, os, shutil, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r02p"), os.path.join(ROOT, "profiles")
CTUS = 510                                                     # 960x544 PMC workload


def db(sub):
    f = sorted(glob.glob(os.path.join(G, sub, "*", "*.db")), key=os.path.getmtime)    # gpurun merges passes into the same directory: the newest
    return sqlite3.connect(f[-1]) if f else None


def json_line(path):
    for l in open(path, errors="replace").read().splitlines():
        if l.startswith("{"):
            return l
    raise SystemExit("no JSON line in " + path)


con = db("prof")
with open(os.path.join(P, "r02_bench_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        mn, mx = con.execute("select min(duration), max(duration) from kernels where name = ?", (name,)).fetchone()
        # top_kernels reports microseconds; kernels.duration nanoseconds
        w.writerow([name, calls, round(total * 1e3), round(avg * 1e3), round(pct, 6), mn, mx])
open(os.path.join(P, "r02_bench_line_under_rocprof.json"), "w").write(json_line(os.path.join(G, "bench_under_rocprof.json")) + "\n")


def counters(sub):
    c = db(sub)
    out = {}
    if c is None:
        return out
    for name, val in c.execute("select counter_name, sum(value) from counters_collection where kernel_name like 'hevce_analyse%' group by counter_name"):
        out[name] = val
    return out


res = {}
for tag, label in (("pmc", "cluster of 4 workgroups (default for one picture)"), ("pmc1", "one workgroup per picture (HEVCE_CLUSTER=1)")):
    v = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE", "INST"):
        v.update(counters(f"{tag}_{c}"))
    if not v:
        continue
    fetch_b, write_b = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024   # KB units; gfx950 reports half of wide reads (MI355X_MICROARCH.md)
    res[tag] = {"config": label, "FETCH_SIZE_KB_raw": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"], "fetch_bytes": fetch_b, "write_bytes": write_b,
                "hbm_bytes_per_ctu": (fetch_b + write_b) / CTUS,
                "valu_insts_per_ctu": v.get("SQ_INSTS_VALU", 0) / CTUS, "salu_insts_per_ctu": v.get("SQ_INSTS_SALU", 0) / CTUS,
                "busy_cu_cycles_per_ctu": v.get("SQ_BUSY_CU_CYCLES", 0) / CTUS,
                "valu_insts_per_busy_cu_cycle": (v.get("SQ_INSTS_VALU", 0) / v["SQ_BUSY_CU_CYCLES"]) if v.get("SQ_BUSY_CU_CYCLES") else None}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CU_CYCLES (separate passes, tools/r02_profile_run.sh) on "
                     "bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --height 544 --width 960 (510 CTUs); FETCH_SIZE doubled per "
                     "MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); a compute unit issues at most 2 wave64 VALU instructions per cycle "
                     "(4 SIMD-32, 2 cycles per wave64 instruction)", **res}, open(os.path.join(P, "r02_pmc_summary.json"), "w"), indent=1)
main = res.get("pmc") or res.get("pmc1")
json.dump({"source": "profiles/r02_pmc_summary.json", "scaled_from": "960x544 (510 CTUs), cluster of 4", "ctus": CTUS, "fetch_bytes": main["fetch_bytes"],
           "write_bytes": main["write_bytes"], "hbm_bytes_per_ctu": main["hbm_bytes_per_ctu"]}, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
if os.path.exists(os.path.join(G, "pytest_gpu.log")):
    shutil.copy(os.path.join(G, "pytest_gpu.log"), os.path.join(P, "r02_pytest_gpu.log"))
print(json.dumps(res, indent=1))
