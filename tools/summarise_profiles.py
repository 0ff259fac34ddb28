#!/usr/bin/env python3
"""Developer tool: fold what tools/collect_profiles.sh left under gpurun_out/<tag>prof/ into profiles/.
    python tools/summarise_profiles.py r02
Writes profiles/<tag>_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the bench command),
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
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", tag + "prof")
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
out = {"command": "rocprofv3 --pmc <one counter group per pass> --output-format csv -- python3 bench.py --steps 2 --warmup 1 "
                  "--cpu-sample 0 --rebuild-every 0   (tools/collect_profiles.sh)",
       "workload": "1024000-atom bcc-Fe", "per_launch_mean": res}
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_pmc_counters.json"), "w"), indent=1, sort_keys=True)
ks = glob.glob(os.path.join(src, "trace", "*kernel_stats.csv"))
if ks:
    shutil.copy(ks[0], os.path.join(ROOT, "profiles", tag + "_bench_kernel_stats.csv"))
bj = os.path.join(src, "bench_under_rocprof.json")
if os.path.exists(bj):
    shutil.copy(bj, os.path.join(ROOT, "profiles", tag + "_bench_under_rocprof.json"))
for k in sorted(res):
    if "fe_" in k or "mlp" in k:
        e = res[k]
        cyc = e.get("GRBM_GUI_ACTIVE", 0) / 8
        print("%-44s valu/atom %6.0f  valu-busy %.2f  lds-busy %.2f  hbm %.2f-%.2f GB  atomics %.3g  mfma_busy_cycles %.3g" % (
            k, e.get("SQ_INSTS_VALU", 0) / 1.024e6, e.get("SQ_ACTIVE_INST_VALU", 0) * 4 / 1024 / cyc if cyc else 0,
            e.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc if cyc else 0,     # the LDS pipe is per CU
            e.get("hbm_bytes_lower", 0) / 1e9, e.get("hbm_bytes_upper", 0) / 1e9, e.get("TCC_EA0_ATOMIC_sum", 0),
            e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)))
