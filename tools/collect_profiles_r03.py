#!/usr/bin/env python3
"""Round 3.  Turn one measurement pass (tools/r03_profile_run.sh on the GPU box -> gpurun_out/r03p/) into the tracked summaries under
profiles/: rocprofv3 writes rocpd databases (sqlite) on this image; the kernel statistics and the PMC values are read from
their `top_kernels` / `kernels` / `counters_collection` views."""
import csv, glob, json, os, shutil, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "r03p"), os.path.join(ROOT, "profiles")
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
with open(os.path.join(P, "r03_bench_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for name, calls, total, avg, pct in con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"):
        mn, mx = con.execute("select min(duration), max(duration) from kernels where name = ?", (name,)).fetchone()
        # top_kernels reports microseconds; kernels.duration nanoseconds
        w.writerow([name, calls, round(total * 1e3), round(avg * 1e3), round(pct, 6), mn, mx])
open(os.path.join(P, "r03_bench_line_under_rocprof.json"), "w").write(json_line(os.path.join(G, "bench_under_rocprof.json")) + "\n")


def counters(sub):
    c = db(sub)
    out = {}
    if c is None:
        return out
    for name, val in c.execute("select counter_name, sum(value) from counters_collection where kernel_name like 'hevce_analyse%' group by counter_name"):
        out[name] = val
    return out


res = {}
for tag, label in (("pmc", "cluster of 7 workgroups (default for one picture)"), ("pmc1", "one workgroup per picture (HEVCE_CLUSTER=1)")):
    v = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE", "INST", "LDS", "LDSW"):
        v.update(counters(f"{tag}_{c}"))
    if not v:
        continue
    fetch_b, write_b = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024   # KB units; gfx950 reports half of wide reads (MI355X_MICROARCH.md)
    res[tag] = {"config": label, "FETCH_SIZE_KB_raw": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"], "fetch_bytes": fetch_b, "write_bytes": write_b,
                "hbm_bytes_per_ctu": (fetch_b + write_b) / CTUS,
                "valu_insts_per_ctu": v.get("SQ_INSTS_VALU", 0) / CTUS, "salu_insts_per_ctu": v.get("SQ_INSTS_SALU", 0) / CTUS,
                "busy_cu_cycles_per_ctu": v.get("SQ_BUSY_CU_CYCLES", 0) / CTUS,
                "valu_insts_per_busy_cu_cycle": (v.get("SQ_INSTS_VALU", 0) / v["SQ_BUSY_CU_CYCLES"]) if v.get("SQ_BUSY_CU_CYCLES") else None,
                # LDS: SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT count cycles of the LDS pipe (indexed operations issuing / stalled by bank conflicts),
                # SQ_WAIT_INST_LDS wave-cycles spent waiting for an LDS instruction (quad-cycles x 4 per the guide's unit note are NOT applied: raw)
                "SQ_LDS_BANK_CONFLICT": v.get("SQ_LDS_BANK_CONFLICT"), "SQ_LDS_IDX_ACTIVE": v.get("SQ_LDS_IDX_ACTIVE"),
                "lds_bank_conflict_fraction_of_active": (v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]) if v.get("SQ_LDS_IDX_ACTIVE") else None,
                "SQ_WAIT_INST_LDS": v.get("SQ_WAIT_INST_LDS"), "SQ_INSTS_LDS": v.get("SQ_INSTS_LDS"), "SQ_WAVE_CYCLES": v.get("SQ_WAVE_CYCLES"),
                "wait_inst_lds_fraction_of_wave_cycles": (v["SQ_WAIT_INST_LDS"] / v["SQ_WAVE_CYCLES"]) if v.get("SQ_WAVE_CYCLES") and v.get("SQ_WAIT_INST_LDS") is not None else None,
                "lds_insts_per_ctu": (v["SQ_INSTS_LDS"] / CTUS) if v.get("SQ_INSTS_LDS") else None}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CU_CYCLES / SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE / SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES (separate passes, tools/r03_profile_run.sh) on "
                     "bench.py --steps 1 --warmup 0 --no-extras --no-cpu-baseline --height 544 --width 960 (510 CTUs); FETCH_SIZE doubled per "
                     "MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); a compute unit issues at most 2 wave64 VALU instructions per cycle "
                     "(4 SIMD-32, 2 cycles per wave64 instruction)", **res}, open(os.path.join(P, "r03_pmc_summary.json"), "w"), indent=1)
main = res.get("pmc") or res.get("pmc1")
json.dump({"source": "profiles/r03_pmc_summary.json", "scaled_from": "960x544 (510 CTUs), cluster of 7", "ctus": CTUS, "fetch_bytes": main["fetch_bytes"],
           "write_bytes": main["write_bytes"], "hbm_bytes_per_ctu": main["hbm_bytes_per_ctu"]}, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
for sub in sorted(os.listdir(G)):                             # the raw per-dispatch counter tables rocprofv3 wrote (-f csv), one file per pass
    for f in glob.glob(os.path.join(G, sub, "*", "*counter_collection.csv")):
        shutil.copy(f, os.path.join(P, f"r03_{sub}_960x544_counter_collection.csv"))
for f in glob.glob(os.path.join(G, "prof", "*", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(P, "r03_bench_kernel_stats_rocprofv3.csv"))
for name in ("bench_960x544.json", "phases.txt"):
    if os.path.exists(os.path.join(G, name)):
        shutil.copy(os.path.join(G, name), os.path.join(P, "r03_" + name))
if os.path.exists(os.path.join(G, "pytest_gpu.log")):
    shutil.copy(os.path.join(G, "pytest_gpu.log"), os.path.join(P, "r03_pytest_gpu.log"))
print(json.dumps(res, indent=1))
