#!/bin/bash
# Developer tool (GPU box): a few counter passes of the default bench, summarised per kernel (sum over dispatches of the counters)
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --list-avail > $out/avail.txt 2>&1
n=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
           "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
    n=$((n+1))
    rocprofv3 --pmc $grp --output-format csv -d $out/pmc$n -o pmc -- python3 bench.py --steps 2 --warmup 1 --cpu-sample 0 --rebuild-every 0 --secondary 0 \
        > $out/pmc$n.json 2> $out/pmc$n.err || echo "pmc pass $n FAILED"
    python3 - $out/pmc$n <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], k)
        if key not in seen: seen.add(key); cnt[k] += 1
for k in acc:
    if "annp_fe" in k or "annp_mlp" in k:
        print(k, cnt[k], {c: round(v / cnt[k]) for c, v in acc[k].items()})
PY
done
