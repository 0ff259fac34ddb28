#!/bin/bash
# Developer tool: one GPU-box session = a list of steps, each under its own timeout, stopping at the first failure.
#   bash tools/gpu_session.sh <name> <step> [<step> ...]      steps: newtests alltests kbench bench hostpath selfwire trace128k rehearse prof profni profanna
set -o pipefail
name=$1; shift
out=gpurun_out/$name
mkdir -p $out
cd $GRAFT_REPO_ROOT
for step in "$@"; do
    echo "=== $step $(date +%T)"
    case $step in
    newtests) timeout -k 10 900 python -m pytest tests/test_gpu_fe_desc_sh.py tests/test_gpu_cpp_md.py tests/test_gpu_step_kernels.py tests/test_gpu_ilist.py tests/test_gpu_hostpath.py tests/test_compat_boundary.py -m gpu -x -q > $out/newtests.log 2>&1 || { tail -40 $out/newtests.log; exit 1; } ; tail -3 $out/newtests.log ;;
    alltests) timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $out/alltests.log 2>&1 || { tail -40 $out/alltests.log; exit 1; } ; tail -3 $out/alltests.log ;;
    kbench)   timeout -k 10 300 python tools/kbench.py ni 40 40 80 > $out/kbench_ni.log 2>&1 || { tail -20 $out/kbench_ni.log; exit 1; }; tail -1 $out/kbench_ni.log
              timeout -k 10 300 python tools/kbench.py anna 80 > $out/kbench_anna.log 2>&1 || { tail -20 $out/kbench_anna.log; exit 1; }; tail -1 $out/kbench_anna.log ;;
    bench)    timeout -k 10 600 python bench.py > $out/bench.json 2> $out/bench.err || { tail -20 $out/bench.err; exit 1; }; cat $out/bench.json ;;
    hostpath) timeout -k 10 600 python tools/hostpath_bench.py 80 > $out/hostpath.log 2>&1 || { tail -20 $out/hostpath.log; exit 1; }; cat $out/hostpath.log ;;
    selfwire) ANNP_FORCE_DIST=1 ANNP_BENCH_WIRE_SELF=1 timeout -k 10 300 python bench.py --cells 40 --steps 10 --secondary 0 --cpu-sample 0 > $out/selfwire.json 2> $out/selfwire.err || { tail -20 $out/selfwire.err; exit 1; }; cat $out/selfwire.json
              ANNP_FORCE_DIST=1 ANNP_BENCH_WIRE_SELF=1 ANNP_BENCH_WIRE=lib timeout -k 10 300 python bench.py --cells 40 --steps 10 --secondary 0 --cpu-sample 0 > $out/selfwire_lib.json 2> $out/selfwire_lib.err || { tail -20 $out/selfwire_lib.err; exit 1; }; cat $out/selfwire_lib.json
              timeout -k 10 300 python bench.py --cells 40 --steps 10 --secondary 0 --cpu-sample 0 > $out/selfwire_ref.json 2> $out/selfwire_ref.err || { tail -20 $out/selfwire_ref.err; exit 1; }; cat $out/selfwire_ref.json ;;
    trace128k) cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
              timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t128k -o t -- python3 bench.py --cells 40 --steps 20 --warmup 5 --cpu-sample 0 --rebuild-every 10 --secondary 0 > $out/b128k.json 2> $out/b128k.err || exit 1
              ANNP_BENCH_TORCH_STEP=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t128k_torch -o t -- python3 bench.py --cells 40 --steps 20 --warmup 5 --cpu-sample 0 --rebuild-every 0 --secondary 0 > $out/b128k_torch.json 2> $out/b128k_torch.err || exit 1 ;;
    rehearse) for n in 2 4; do ANNP_BENCH_SHARE_GPU=1 ANNP_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus $n --cells 40 --steps 10 --warmup 2 --cpu-sample 0 --secondary 0 > $out/share$n.json 2> $out/share$n.err || { tail -20 $out/share$n.err; exit 1; }; done
              timeout -k 10 300 python bench.py --cells 40 --steps 10 --warmup 2 --cpu-sample 0 --secondary 0 > $out/share1.json 2> $out/share1.err || exit 1 ;;
    prof)     timeout -k 10 1100 bash tools/collect_profiles.sh $name || exit 1 ;;
    profni)   timeout -k 10 900 bash tools/collect_profiles.sh $name ni || exit 1 ;;
    profanna) timeout -k 10 900 bash tools/collect_profiles.sh $name anna || exit 1 ;;
    *) echo "unknown step $step"; exit 2 ;;
    esac
done
echo "=== done $(date +%T)"
