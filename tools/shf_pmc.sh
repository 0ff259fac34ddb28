#!/bin/bash
# Developer tool (GPU box): counter passes of the Chebyshev force pass, this round's kernel and round 4's, one table.
#   bash tools/shp_pmc.sh <name>
out=gpurun_out/$1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for variant in ${VARIANTS:-r5}; do
  if [ $variant != r5 ]; then export ANNP_HIP_LIBRARY=$PWD/meng_zhang_amd/libannp_hip_$variant.so; else unset ANNP_HIP_LIBRARY; fi
  n=0
  for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
             "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" \
             "SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
             "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_ACTIVE_INST_EXP_GDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
    n=$((n+1))
    timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $out/${variant}_pmc$n -o pmc -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --rebuild-every 0 --secondary 0 \
        > $out/${variant}_pmc$n.json 2> $out/${variant}_pmc$n.err || { echo "pass $variant $n failed"; tail -5 $out/${variant}_pmc$n.err; exit 1; }
  done
done
python3 - $out ${VARIANTS:-r5} <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
for variant in sys.argv[2:] or ["r5"]:
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("%s/%s_pmc*/**/*counter_collection.csv" % (out, variant), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "annp_fe_force_sh" in k or "annp_fe_desc_sh" in k:
                res[k.split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in res.items():
        e = {c: sum(x) / len(x) for c, x in v.items()}
        cyc = e.get("GRBM_GUI_ACTIVE", 0) / 8
        print(variant, k[:50])
        print("   " + "  ".join("%s %.4g" % (c, e[c]) for c in sorted(e) if c.startswith("SQC") or c in ("SQ_IFETCH", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_FLAT", "SQ_INSTS_SMEM", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU")))
        print("   valu/atom %.0f  salu/atom %.0f  lds/atom %.0f  vmem_rd/atom %.1f vmem_wr/atom %.1f  cycles %.3g  valu issue %.3f  any-inst active %.3f  waves/SIMD %.2f  lds pipe %.3f  bank conflicts %.3f  wait_inst_any %.3g wait_lds %.3g" % (
            e.get("SQ_INSTS_VALU", 0) / 1024000, e.get("SQ_INSTS_SALU", 0) / 1024000, e.get("SQ_INSTS_LDS", 0) / 1024000, e.get("SQ_INSTS_VMEM_RD", 0) / 1024000, e.get("SQ_INSTS_VMEM_WR", 0) / 1024000, cyc,
            e.get("SQ_INSTS_VALU", 0) * 4 / 1024 / cyc if cyc else 0, e.get("SQ_ACTIVE_INST_ANY", 0) * 4 / 1024 / cyc if cyc else 0, e.get("SQ_WAVE_CYCLES", 0) * 4 / 1024 / cyc if cyc else 0,
            e.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc if cyc else 0, e.get("SQ_LDS_BANK_CONFLICT", 0) / max(e.get("SQ_LDS_IDX_ACTIVE", 1), 1), e.get("SQ_WAIT_INST_ANY", 0), e.get("SQ_WAIT_INST_LDS", 0)))
PY
