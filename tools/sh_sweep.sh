#!/bin/bash
# Developer tool (GPU box): descriptor-pass time of the default bench for a few launch shapes of annp_fe_desc_sh
out=gpurun_out/$1; mkdir -p $out
for cfg in "0 x" "0 y" "pairs"; do
    set -- $cfg
    if [ "$1" = "pairs" ]; then export ANNP_HIP_FE_DESC=pairs ANNP_HIP_FE_FORCE=pairs; else export ANNP_HIP_SH_WPB=$1; fi
    timeout -k 10 200 python bench.py --steps 5 --secondary 0 --cpu-sample 0 --rebuild-every 0 > $out/b.json 2> $out/b.err || { tail -5 $out/b.err; exit 1; }
    python -c "import sys,json; d=json.load(open('$out/b.json')); print('$cfg', round(d['value']/1e6,2), d['kernel_ms'])"
done
