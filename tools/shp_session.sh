#!/bin/bash
# Developer tool (GPU box): parity of the Chebyshev force pass, then kernel timings, new kernel against round 4's.
set -o pipefail
out=gpurun_out/$1; shift
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_async.py -m gpu -x -q > $out/parity.log 2>&1 || { tail -40 $out/parity.log; exit 1; }
tail -3 $out/parity.log
for run in "$@"; do
  ANNP_HIP_FE_FORCE=walk ANNP_HIP_SHP_RUN=$run timeout -k 10 200 python tools/kbench.py fe 80 > $out/k_run$run.log 2>&1 || { tail -20 $out/k_run$run.log; exit 1; }
  echo "run $run: $(grep atoms= $out/k_run$run.log)"
  ANNP_HIP_FE_FORCE=walk ANNP_HIP_SHP_ROLES=number ANNP_HIP_SHP_RUN=$run timeout -k 10 200 python tools/kbench.py fe 80 > $out/k_run${run}n.log 2>&1 || { tail -20 $out/k_run${run}n.log; exit 1; }
  echo "run $run, roles by number: $(grep atoms= $out/k_run${run}n.log)"
done
ANNP_HIP_FE_FORCE=sh timeout -k 10 200 python tools/kbench.py fe 80 > $out/k_r4.log 2>&1 || { tail -20 $out/k_r4.log; exit 1; }
echo "r4: $(grep atoms= $out/k_r4.log)"
if [ -f meng_zhang_amd/libannp_hip_stamps.so ]; then
  ANNP_HIP_FE_FORCE=walk ANNP_HIP_LIBRARY=$PWD/meng_zhang_amd/libannp_hip_stamps.so ANNP_HIP_SHP_RUN=16 timeout -k 10 300 python tools/shp_stamps.py 80 > $out/stamps.log 2>&1 || { tail -20 $out/stamps.log; exit 1; }
  grep -v amdgpu.ids $out/stamps.log | head -60
fi
