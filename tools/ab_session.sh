#!/bin/bash
# Developer tool: A/B timing of kernel variants on ONE box (the boxes of the pool differ by +-3 %, more than most steps are worth):
# tools/kbench.py under each of the given environments in turn, `rounds` times over.
#   bash tools/ab_session.sh <name> <rounds> "<kbench args>" "<env A>" "<env B>" ...       an environment: VAR=value words, or "-" for none;
#   LIB=<file> in an environment selects meng_zhang_amd/<file> as ANNP_HIP_LIBRARY
#   TESTS="tests/test_gpu_fe_desc_sh.py ..." in the caller's environment: run first (under the first TESTS_ENVS environments, default all), stop if red
set -o pipefail
name=$1; rounds=$2; kargs=$3; shift 3
out=gpurun_out/$name
mkdir -p $out
cd $GRAFT_REPO_ROOT
expand() { local e=""; for w in $1; do case $w in -) ;; LIB=*) e="$e ANNP_HIP_LIBRARY=$GRAFT_REPO_ROOT/meng_zhang_amd/${w#LIB=}" ;; *) e="$e $w" ;; esac; done; echo $e; }
if [ -n "$TESTS" ]; then
    k=0
    for env in "$@"; do
        [ $k -ge ${TESTS_ENVS:-99} ] && break
        echo "=== tests under [$env] $(date +%T)"
        env $(expand "$env") timeout -k 10 600 python -m pytest $TESTS -m gpu -x -q > $out/tests_$k.log 2>&1 || { tail -30 $out/tests_$k.log; exit 1; }
        tail -1 $out/tests_$k.log
        k=$((k + 1))
    done
fi
for r in $(seq 1 $rounds); do
    k=0
    for env in "$@"; do
        env $(expand "$env") timeout -k 10 300 python tools/kbench.py $kargs > $out/kb_${k}_$r.log 2>&1 || { tail -20 $out/kb_${k}_$r.log; exit 1; }
        echo "[$env] $(grep -E 'atoms=' $out/kb_${k}_$r.log)"
        k=$((k + 1))
    done
done
echo "=== done $(date +%T)"
