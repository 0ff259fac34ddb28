#!/usr/bin/env python3
"""Developer tool: copy what a `tools/gpu_session.sh <name> ...` run left under gpurun_out/<name>/ into profiles/ (tracked).
   python tools/gather_evidence.py r03 final      # tag, session name"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
src, dst = os.path.join(ROOT, "gpurun_out", name), os.path.join(ROOT, "profiles")


def load(f):
    p = os.path.join(src, f)
    return json.load(open(p)) if os.path.exists(p) and os.path.getsize(p) else None


if load("bench.json"):
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, tag + "_bench_line.json"))
for f, out in (("hostpath.log", "_hostpath.txt"), ("kbench_ni.log", "_kbench_ni.txt"), ("kbench_anna.log", "_kbench_anna.txt")):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, tag + out))
wires = {k: load(f) for k, f in (("torch.distributed batch_isend_irecv (ANNP_BENCH_WIRE_SELF=1)", "selfwire.json"),
                                 ("library annp_hip_comm_route (ANNP_BENCH_WIRE_SELF=1 ANNP_BENCH_WIRE=lib)", "selfwire_lib.json"),
                                 ("local copies (no wire)", "selfwire_ref.json"))}
if all(wires.values()):
    json.dump({"what": "one rank as its own x neighbour: the x-periodic images travel through RCCL send/recv to the same rank "
                       "(bench.py --cells 40 --steps 10 --secondary 0 --cpu-sample 0, ANNP_FORCE_DIST=1)",
               "runs": {k: {"ms_per_step": v["ms_per_step"], "energy_per_atom_eV": v["energy_per_atom_eV"], "value": v["value"],
                            "mini_md": v["mini_md"]["value"], "backend": v["config"]["backend"], "wire": v["config"].get("wire"),
                            "halo_bytes_per_step": v["config"]["halo_bytes_per_step"]} for k, v in wires.items()}},
              open(os.path.join(dst, tag + "_rccl_self_wire.json"), "w"), indent=1)
share = {n: load("share%d.json" % n) for n in (1, 2, 4)}
if all(share.values()):
    json.dump({"what": "N ranks of bench.py sharing ONE MI355X over gloo with the halo bounced through the host (ANNP_BENCH_SHARE_GPU=1 "
                       "ANNP_BENCH_BACKEND=gloo): a rehearsal of the N > 1 control flow with the library's step kernels, never a measurement",
               "runs": {str(n): {"atoms_rank": v["config"]["atoms_rank"], "ghosts_rank": v["config"]["ghosts_rank"],
                                 "energy_per_atom_eV": v["energy_per_atom_eV"], "ms_per_step": v["ms_per_step"],
                                 "atoms_that_changed_rank_in_mini_md": v["mini_md"]["atoms_that_changed_rank"]} for n, v in share.items()}},
              open(os.path.join(dst, tag + "_rehearsal_ranks_sharing_one_gpu.json"), "w"), indent=1)
for sub, out in (("t128k", "_step_128k_kernel_stats.csv"), ("t128k_torch", "_step_128k_torch_ops_kernel_stats.csv")):
    f = os.path.join(src, sub, "t_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, tag + out))
        rows = [r for r in csv.DictReader(open(f)) if int(r["Calls"]) >= 20]
        print(sub, json.load(open(os.path.join(src, "b" + sub[1:] + ".json")))["ms_per_step"], "ms per step")
        for r in rows:
            print("   %-60s calls %4s  avg %8.2f us" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
