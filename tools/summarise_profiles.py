#!/usr/bin/env python3
"""Developer tool: fold what tools/collect_profiles.sh left under gpurun_out/<tag>prof/ into profiles/.
    python tools/summarise_profiles.py r03 [ni|anna]
Writes profiles/<tag>[_<workload>]_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the bench command),
profiles/<tag>_bench_under_rocprof.json (the line bench.py printed in that run) and
profiles/<tag>_pmc_counters.json (per-kernel means of the counter passes, HBM bytes per launch from FETCH_SIZE and
WRITE_SIZE as MI355X_MICROARCH.md prescribes: separate passes, KB units, FETCH_SIZE doubled on gfx950 = upper bound)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
wl = sys.argv[2] if len(sys.argv) > 2 else "fe"
suffix = "" if wl == "fe" else "_" + wl
src = os.path.join(ROOT, "gpurun_out", tag + "prof" + suffix)
natoms = {"fe": 1024000, "anna": 1024000, "ni": 512000}[wl]
res = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(src, "pmc*", "*counter_collection.csv"))):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if "annp" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        for c, x in v.items():
            res[k][c] = sum(x) / len(x)
            res[k]["launches_" + c] = len(x)
for k, e in res.items():
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["hbm_bytes_lower"] = (e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
        e["hbm_bytes_upper"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
out = {"command": "rocprofv3 --pmc <one counter group per pass> --output-format csv -- python3 bench.py%s --steps 2 --warmup 1 "
                  "--cpu-sample 0 --rebuild-every 0 --secondary 0   (tools/collect_profiles.sh)" % ("" if wl == "fe" else " --workload " + wl),
       "workload": {"fe": "1024000-atom bcc-Fe", "ni": "512000-atom fcc-Ni", "anna": "1024000-atom bcc-Fe, pair_style anna_adp"}[wl],
       "per_launch_mean": res}
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
ks = glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))
# average launch duration of every kernel from the kernel trace of the same command, and what follows from it together with the
# counters: the shader clock the chip held under that kernel (GRBM_GUI_ACTIVE is summed over the 8 XCDs), HBM bytes per second,
# FP64-MFMA flop per second (SQ_INSTS_VALU_MFMA_MOPS_F64 counts 512 flop each: v_mfma_f64_16x16x4 = 2048 flop = 4 of them)
if ks:
    for r in csv.DictReader(open(ks[0])):
        k = r["Name"].split("(")[0].replace("void ", "")
        if k in res:
            e = res[k]
            e["avg_ms"] = float(r["AverageNs"]) / 1e6
            e["trace_calls"] = int(r["Calls"])
            if e.get("GRBM_GUI_ACTIVE"):
                e["clock_GHz"] = e["GRBM_GUI_ACTIVE"] / 8 / float(r["AverageNs"])
            if "hbm_bytes_upper" in e:
                e["hbm_GBps_upper"] = e["hbm_bytes_upper"] / float(r["AverageNs"])
                e["hbm_GBps_lower"] = e["hbm_bytes_lower"] / float(r["AverageNs"])
            if e.get("SQ_INSTS_VALU_MFMA_MOPS_F64"):
                e["mfma_TFLOPs"] = e["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512 / float(r["AverageNs"]) / 1e3
                e["mfma_busy_share"] = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * e["GRBM_GUI_ACTIVE"] / 8) if e.get("GRBM_GUI_ACTIVE") else None
json.dump(out, open(os.path.join(ROOT, "profiles", tag + suffix + "_pmc_counters.json"), "w"), indent=1, sort_keys=True)
if ks:
    shutil.copy(ks[0], os.path.join(ROOT, "profiles", tag + suffix + "_bench_kernel_stats.csv"))
bj = os.path.join(src, "bench_under_rocprof.json")
if os.path.exists(bj):
    shutil.copy(bj, os.path.join(ROOT, "profiles", tag + suffix + "_bench_under_rocprof.json"))
for k in sorted(res):
    if any(t in k for t in ("fe_", "mlp", "ni_", "anna")):
        e = res[k]
        cyc = e.get("GRBM_GUI_ACTIVE", 0) / 8
        # SQ_WAVE_CYCLES counts quad-cycles summed over waves: / (cycles / 4) / 1024 SIMDs = resident waves per SIMD
        print("%-60s valu/atom %6.0f  valu-busy %.2f  waves/SIMD %.2f  lds-busy %.2f  hbm %.2f-%.2f GB  atomics %.3g  mfma_busy_cycles %.3g" % (
            k[:60], e.get("SQ_INSTS_VALU", 0) / natoms, e.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / cyc if cyc else 0,
            e.get("SQ_WAVE_CYCLES", 0) * 4 / 1024 / cyc if cyc else 0,
            e.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc if cyc else 0,     # the LDS pipe is per CU
            e.get("hbm_bytes_lower", 0) / 1e9, e.get("hbm_bytes_upper", 0) / 1e9, e.get("TCC_EA0_ATOMIC_sum", 0),
            e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)))
