#!/usr/bin/env python3
"""Developer tool: fold the rocprofv3 outputs gpurun merged under gpurun_out/ into profiles/ (round tag as argv[1]).
Expects: gpurun_out/prof_<tag>/ (kernel-trace --stats), gpurun_out/pmcd_*/ (one directory per --pmc pass)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
prof = sys.argv[2] if len(sys.argv) > 2 else "prof_r01d"
res = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "pmcd_*"))):
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if "annp" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            for c, x in v.items():
                res[k][c] = sum(x) / len(x)
for k, e in res.items():
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        # FETCH_SIZE/WRITE_SIZE are in KB; FETCH_SIZE reads half the bytes of wide coalesced reads on gfx950
        # (MI355X_MICROARCH.md, HBM): both bounds are kept
        e["hbm_bytes_lower"] = (e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
        e["hbm_bytes_upper"] = (2 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
out = {"command": "rocprofv3 --pmc <one counter group per pass> --output-format csv -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0",
       "workload": "1024000-atom bcc-Fe", "per_launch_mean": res}
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_pmc_counters.json"), "w"), indent=1, sort_keys=True)
ks = glob.glob(os.path.join(ROOT, "gpurun_out", prof, "*", "*kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(ROOT, "profiles", tag + "_bench_kernel_stats.csv"))
bj = os.path.join(ROOT, "gpurun_out", prof + "_bench.json")
if os.path.exists(bj):
    shutil.copy(bj, os.path.join(ROOT, "profiles", tag + "_bench_under_rocprof.json"))
for k in sorted(res):
    if "fe_" in k or "mlp" in k:
        e = res[k]
        cyc = e.get("GRBM_GUI_ACTIVE", 0) / 8
        print("%-36s valu/atom %6.0f  busy %.2f  hbm %.2f-%.2f GB  mfma_busy_cycles %.3g" % (
            k, e.get("SQ_INSTS_VALU", 0) / 1.024e6, e.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / cyc if cyc else 0,
            e.get("hbm_bytes_lower", 0) / 1e9, e.get("hbm_bytes_upper", 0) / 1e9, e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)))
