#!/usr/bin/env python3
"""Copy the evidence of one measurement pass from gpurun_out/ (scratch) into profiles/ (tracked) and derive the summaries.
usage: collect_profiles.py <tag> [round]     e.g. collect_profiles.py r01h r01
Expects the files the round's measurement command writes: bench_<tag>.json, bench_<tag>_rocprof.json, prof_<tag>/,
pmc_<tag>_<counters>/ (FETCH_SIZE, WRITE_SIZE, SQ_LDS_BANK_CO.., SQ_INSTS_VALU_.., SQ_ACTIVE_INST.., SQ_INST_CYCLES..),
batch_<tag>.json, tiles_<tag>.json, fullsize_parity_*.json, pytest.log."""
import csv, glob, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r01")
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
CTUS = 510                                                     # 960x544 PMC workload


def json_line(path):
    for l in open(path, errors="replace").read().splitlines():
        if l.startswith("{"):
            return l
    raise SystemExit("no JSON line in " + path)


def counters(pattern, dst):
    f = glob.glob(os.path.join(G, pattern, "*", "*counter_collection.csv"))
    if not f:
        return {}
    shutil.copy(f[0], os.path.join(P, dst))
    v = {}
    for r in csv.DictReader(open(f[0])):
        if "hevce_analyse" in r["Kernel_Name"]:
            v[r["Counter_Name"]] = v.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return v


open(os.path.join(P, f"{rnd}_bench_line.json"), "w").write(json_line(os.path.join(G, f"bench_{tag}.json")) + "\n")
open(os.path.join(P, f"{rnd}_bench_line_under_rocprof.json"), "w").write(json_line(os.path.join(G, f"bench_{tag}_rocprof.json")) + "\n")
for k in ("kernel_stats", "domain_stats"):
    shutil.copy(glob.glob(os.path.join(G, f"prof_{tag}", "*", f"*{k}.csv"))[0], os.path.join(P, f"{rnd}_bench_{k}.csv"))
v = {}
v.update(counters(f"pmc_{tag}_FETCH_SIZE", f"{rnd}_pmc_FETCH_SI_960x544.csv"))
v.update(counters(f"pmc_{tag}_WRITE_SIZE", f"{rnd}_pmc_WRITE_SI_960x544.csv"))
v.update(counters(f"pmc_{tag}_SQ_LDS_BANK*", f"{rnd}_pmc_SQ_LDS_B_960x544.csv"))
fetch_b, write_b = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024   # KB units; gfx950 reports half of wide reads (MI355X_MICROARCH.md)
per = (fetch_b + write_b) / CTUS
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on bench.py --no-cpu-baseline --warmup 0 --height 544 "
                     "--width 960; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads); KB units",
           "ctus": CTUS, "fetch_bytes": fetch_b, "write_bytes": write_b, "hbm_bytes_per_ctu": per},
          open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
json.dump({"run": f"bench.py --no-cpu-baseline --warmup 0 --height 544 --width 960, {CTUS} CTUs, pass {tag}",
           "FETCH_SIZE_KB_raw": v["FETCH_SIZE"], "WRITE_SIZE_KB": v["WRITE_SIZE"], "hbm_bytes_per_ctu": per,
           **{k: v[k] for k in ("SQ_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")},
           "lds_bank_conflict_share": v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], "wait_any_share": v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]},
          open(os.path.join(P, f"{rnd}_pmc_summary.json"), "w"), indent=1)
m = {}
m.update(counters(f"pmc_{tag}_SQ_INSTS_VALU*", f"{rnd}_pmc_INSTS_960x544.csv"))
m.update(counters(f"pmc_{tag}_SQ_ACTIVE_INST*", f"{rnd}_pmc_ACTIVE_960x544.csv"))
m.update(counters(f"pmc_{tag}_SQ_INST_CYCLES*", f"{rnd}_pmc_CYCLES_960x544.csv"))
if m:
    json.dump({"run": f"same command, three more --pmc passes ({tag}): ONE workgroup of 8 wavefronts on one compute unit", **m,
               "valu_insts_per_ctu": m["SQ_INSTS_VALU"] / CTUS, "salu_insts_per_ctu": m["SQ_INSTS_SALU"] / CTUS,
               "lds_insts_per_ctu": m["SQ_INSTS_LDS"] / CTUS, "cu_cycles_per_ctu": m["SQ_BUSY_CU_CYCLES"] / CTUS,
               "valu_issue_utilisation_of_the_cu": m["SQ_INSTS_VALU"] / m["SQ_BUSY_CU_CYCLES"],
               "active_lanes_per_valu_inst": m["SQ_THREAD_CYCLES_VALU"] / m["SQ_INSTS_VALU"],
               "note": "a compute unit issues at most one wave64 VALU instruction per cycle (4 SIMDs x 1 per 4 cycles)"},
              open(os.path.join(P, f"{rnd}_pmc_instmix.json"), "w"), indent=1)
for src, dst in ((f"batch_{tag}.json", f"{rnd}_batch_256x1080p.json"), (f"tiles_{tag}.json", f"{rnd}_tiles_16384x16384.json")):
    if os.path.exists(os.path.join(G, src)):
        open(os.path.join(P, dst), "w").write(json_line(os.path.join(G, src)) + "\n")
fp = sorted(glob.glob(os.path.join(G, "fullsize_parity_*.json")), key=os.path.getmtime)
if fp:
    open(os.path.join(P, f"{rnd}_fullsize_parity_4k.json"), "w").write(json_line(fp[-1]) + "\n")
if os.path.exists(os.path.join(G, "pytest.log")):
    shutil.copy(os.path.join(G, "pytest.log"), os.path.join(P, f"{rnd}_pytest_gpu.log"))
print(json.dumps({"hbm_bytes_per_ctu": per, "valu_util": m.get("SQ_INSTS_VALU", 0) / m.get("SQ_BUSY_CU_CYCLES", 1) if m else None}))
