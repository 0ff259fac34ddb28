#!/bin/bash
# Developer tool, run on the GPU box (through gpurun): rocprofv3 passes of the bench command for profiles/.
#   bash tools/collect_profiles.sh r03            # the metric's workload (default bench.py command)
#   bash tools/collect_profiles.sh r03 ni         # bench.py --workload ni (BASELINE.json config 5)
#   bash tools/collect_profiles.sh r03 anna
# One pass for the kernel trace, then one pass per counter group (never --pmc together with a trace domain; FETCH_SIZE and
# WRITE_SIZE do not fit one pass: MI355X_MICROARCH.md, "rocprofv3 PMC slots").  The program itself stands after `--`.
set -e
tag=${1:-r03}
wl=${2:-fe}
suffix=""
wlargs=""
if [ "$wl" != "fe" ]; then suffix="_$wl"; wlargs="--workload $wl"; fi
out=gpurun_out/${tag}prof${suffix}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 bench.py $wlargs --steps 20 --warmup 5 --cpu-sample 0 --secondary 0 \
    > $out/bench_under_rocprof.json 2> $out/trace.err
echo "trace done"
n=0
failed=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_ATOMIC_sum" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64" \
           "SQ_WAIT_ANY SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
    n=$((n+1))
    rocprofv3 --pmc $grp --output-format csv -d $out/pmc$n -o pmc -- python3 bench.py $wlargs --steps 2 --warmup 1 --cpu-sample 0 --rebuild-every 0 --secondary 0 \
        > $out/pmc$n.json 2> $out/pmc$n.err || { echo "pmc pass $n FAILED (see $out/pmc$n.err)"; tail -5 $out/pmc$n.err; failed=$((failed+1)); }
    echo "pmc pass $n done: $grp"
done
if [ $failed -ne 0 ]; then echo "$failed counter pass(es) failed: nothing of this run goes into profiles/ before that is understood"; exit 1; fi
