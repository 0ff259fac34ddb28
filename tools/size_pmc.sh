#!/bin/bash
# Developer tool (GPU box): clock, issue share and waves of the Chebyshev passes at a small and at the full size.
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cells in 40 80; do
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $out/c$cells -o pmc -- python3 bench.py --cells $cells --steps 4 --warmup 2 --cpu-sample 0 --rebuild-every 0 --secondary 0 > $out/c$cells.json 2> $out/c$cells.err || { tail -5 $out/c$cells.err; exit 1; }
done
python3 - $out <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
for cells in (40, 80):
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    cols = None
    for f in glob.glob("%s/c%d/**/*counter_collection.csv" % (out, cells), recursive=True):
        rd = csv.DictReader(open(f))
        for r in rd:
            cols = list(r.keys())
            k = r["Kernel_Name"]
            if "annp_fe_force_sh" in k or "annp_fe_desc_sh" in k:
                res[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if "Start_Timestamp" in r:
                    res[k.split("(")[0].replace("void ", "")]["_dur"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    print("cells", cells, "columns", cols)
    n = 2 * cells ** 3
    for k, v in res.items():
        e = {c: sum(x) / len(x) for c, x in v.items()}
        cyc = e.get("GRBM_GUI_ACTIVE", 0) / 8
        print("  %-40s cycles %.3g  dur_ns %.4g  clock %.3f GHz  valu/atom %.0f  issue %.3f  waves/SIMD %.2f  cycles/atom %.2f" % (
            k[:40], cyc, e.get("_dur", 0), cyc / e["_dur"] if e.get("_dur") else 0, e.get("SQ_INSTS_VALU", 0) / n, e.get("SQ_INSTS_VALU", 0) * 4 / 1024 / cyc if cyc else 0,
            e.get("SQ_WAVE_CYCLES", 0) * 4 / 1024 / cyc if cyc else 0, cyc / n * 1024))
PY
