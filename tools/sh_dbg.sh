#!/bin/bash
# Developer tool (GPU box): force-pass time of the default bench with parts of annp_fe_force_sh switched off (ANNP_HIP_DBG)
out=gpurun_out/$1; mkdir -p $out
for d in 0 1 2 3; do
    ANNP_HIP_DBG=$d timeout -k 10 200 python bench.py --steps 5 --secondary 0 --cpu-sample 0 --rebuild-every 0 > $out/b.json 2> $out/b.err || { tail -5 $out/b.err; exit 1; }
    python -c "import sys,json; d=json.load(open('$out/b.json')); print('dbg $d', round(d['value']/1e6,2), d['kernel_ms'])"
done
