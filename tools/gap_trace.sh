#!/bin/bash
# Developer tool (GPU box): where a 128 000-atom step's time goes between its kernels (kernel trace of bench.py --cells 40).
out=gpurun_out/$1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -o t -- python3 bench.py --cells ${2:-40} --steps 20 --warmup 5 --cpu-sample 0 --rebuild-every 0 --secondary 0 > $out/b.json 2> $out/b.err || { tail -5 $out/b.err; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:46]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
# the timed region: the last 20 steps = the last 20 force_sh launches
idx = [i for i, r in enumerate(rows) if "annp_fe_force_sh" in r[2]]
first = idx[-20]
# start of that step = the verlet_half before it: walk back to previous force_sh end
start_i = idx[-21] + 1
seg = rows[start_i:]
last_force = idx[-1]
seg = rows[start_i:last_force + 1]
busy = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - seg[0][0]
print("kernels in 20 steps: %d  span %.3f ms  busy %.3f ms  gaps %.3f ms (%.1f %%) per step: span %.4f busy %.4f" % (len(seg), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span, span / 20e6, busy / 20e6))
acc = collections.defaultdict(lambda: [0, 0, 0])
for i, (s, e, n) in enumerate(seg):
    a = acc[n]; a[0] += 1; a[1] += e - s
    if i + 1 < len(seg): a[2] += max(0, seg[i + 1][0] - e)      # gap behind this kernel
for n, (c, d, g) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("  %-46s x%-4d dur %8.1f us/step   gap behind %6.1f us/step" % (n, c, d / 20e3, g / 20e3))
PY
