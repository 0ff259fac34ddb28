#!/bin/bash
# Developer tool (GPU box): kernel timings of the Behler (Ni) passes, the default library first, then every library given
# (developer builds of the same code: -DNI_TSLOTS_N=64, -DNI_RUN_GROUPS=8, -DNI_KEEP_RECORDS=0 ...; the parity tests run on the default library only:
# NI_SKIP_PARITY=1 skips them).
#   bash tools/ni_session.sh <name> [library ...]
set -o pipefail
out=gpurun_out/$1; shift
mkdir -p $out
cd $GRAFT_REPO_ROOT
if [ -z "$NI_SKIP_PARITY" ]; then
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_async.py tests/test_gpu_shapes.py tests/test_golden_vectors.py -m gpu -x -q -k "ni or Ni or behler or golden" > $out/parity.log 2>&1 || { grep -a "^FAILED\|^E  " $out/parity.log | head -20; tail -5 $out/parity.log; exit 1; }
tail -2 $out/parity.log
fi
timeout -k 10 200 python tools/kbench.py ni 40 40 80 > $out/k_ni.log 2>&1 || { tail -20 $out/k_ni.log; exit 1; }
echo "default: $(grep atoms= $out/k_ni.log)"
for lib in "$@"; do
  n=$(basename $lib .so)
  ANNP_HIP_LIBRARY=$PWD/$lib timeout -k 10 200 python tools/kbench.py ni 40 40 80 > $out/k_$n.log 2>&1 || { tail -20 $out/k_$n.log; exit 1; }
  echo "$n: $(grep atoms= $out/k_$n.log)"
done
timeout -k 10 200 python tools/kbench.py ni 40 40 80 > $out/k_ni2.log 2>&1 || { tail -20 $out/k_ni2.log; exit 1; }
echo "default again: $(grep atoms= $out/k_ni2.log)"
