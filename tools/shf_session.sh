#!/bin/bash
# Developer tool (GPU box): parity of the Chebyshev force pass, then kernel timings of its variants.
#   bash tools/shf_session.sh <name> [library ...]     every library given (a developer build of the same code) is timed after the default one
set -o pipefail
out=gpurun_out/$1; shift
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_async.py -m gpu -q > $out/parity.log 2>&1 || { grep -a "^FAILED\|^E  " $out/parity.log | head -20; }
tail -3 $out/parity.log
timeout -k 10 200 python tools/kbench.py fe 80 > $out/k_sh.log 2>&1 || { tail -20 $out/k_sh.log; exit 1; }
echo "places by SIMD: $(grep atoms= $out/k_sh.log)"
ANNP_HIP_SHF_PLACES=number timeout -k 10 200 python tools/kbench.py fe 80 > $out/k_shn.log 2>&1 || { tail -20 $out/k_shn.log; exit 1; }
echo "places by number: $(grep atoms= $out/k_shn.log)"
timeout -k 10 200 python tools/kbench.py fe 40 > $out/k_sh40.log 2>&1 || { tail -20 $out/k_sh40.log; exit 1; }
echo "128 000 atoms: $(grep atoms= $out/k_sh40.log)"
KBENCH_ORDER=lammps timeout -k 10 200 python tools/kbench.py fe 80 > $out/k_lmp.log 2>&1 || { tail -20 $out/k_lmp.log; exit 1; }
echo "atom_modify sort order: $(grep atoms= $out/k_lmp.log)"
for lib in "$@"; do
  n=$(basename $lib .so)
  ANNP_HIP_LIBRARY=$PWD/$lib timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "2000 or random" > $out/parity_$n.log 2>&1 || { tail -30 $out/parity_$n.log; exit 1; }
  ANNP_HIP_LIBRARY=$PWD/$lib timeout -k 10 200 python tools/kbench.py fe 80 > $out/k_$n.log 2>&1 || { tail -20 $out/k_$n.log; exit 1; }
  echo "$n: $(tail -1 $out/parity_$n.log) $(grep atoms= $out/k_$n.log)"
  ANNP_HIP_LIBRARY=$PWD/$lib timeout -k 10 200 python tools/kbench.py fe 40 > $out/k40_$n.log 2>&1 || { tail -20 $out/k40_$n.log; exit 1; }
  echo "$n, 128 000 atoms: $(grep atoms= $out/k40_$n.log)"
  KBENCH_ORDER=lammps ANNP_HIP_LIBRARY=$PWD/$lib timeout -k 10 200 python tools/kbench.py fe 80 > $out/klmp_$n.log 2>&1 || { tail -20 $out/klmp_$n.log; exit 1; }
  echo "$n, atom_modify sort order: $(grep atoms= $out/klmp_$n.log)"
done
