#!/usr/bin/env python3
"""Developer tool: kernel timings (HIP events) of one force evaluation for a synthetic box.
   python tools/kbench.py fe 80      # bcc Fe, 80^3 cells
   python tools/kbench.py ni 40 40 80  # fcc Ni 40x40x80 cells (512 000 atoms)
   python tools/kbench.py anna 80      # bcc Fe, pair_style anna_adp (list cutoff 5.055 + 2 A)
   KBENCH_VIRIAL=1 python tools/kbench.py fe 80     # with the global virial tallied (vflag_global of an NPT step)
   KBENCH_SHUFFLE=1 python tools/kbench.py fe 40    # atoms in random order (no locality of index: the slow path of the force tables)
   KBENCH_ORDER=lammps python tools/kbench.py fe 80 # atoms in the order LAMMPS' atom_modify sort leaves them (bins of 4.25 A, x fastest, random inside a bin)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from annp_testlib import A_FE, A_NI, ANNA_POT, FE_POT, NI_POT, bcc, fcc, perturb
    from meng_zhang_amd import PairANNP
    from meng_zhang_amd.domain import SlabDomain
    from meng_zhang_amd.lib import load_library
    kind = sys.argv[1]
    dims = [int(v) for v in sys.argv[2:5]]
    while len(dims) < 3:
        dims.append(dims[-1])
    reps = 5
    style, rc_list = "annp", 8.5
    if kind == "fe":
        x0, box = bcc(*dims, A_FE)
        pot, el = FE_POT, "Fe"
    elif kind == "anna":
        x0, box = bcc(*dims, A_FE)
        pot, el, style, rc_list = ANNA_POT, "Fe", "anna_adp", 7.055
    else:
        x0, box = fcc(*dims, A_NI)
        pot, el = NI_POT, "Ni"
    xg = perturb(x0, 12345, 0.05)
    if os.environ.get("KBENCH_ORDER") == "lammps":          # what a LAMMPS caller delivers between two sorts
        from meng_zhang_amd.workloads import lammps_sort_order
        xg = xg[lammps_sort_order(xg, box)]
    if os.environ.get("KBENCH_SHUFFLE"):          # atoms in random order: what the force tables do for a caller that does not sort its atoms
        xg = xg[np.random.default_rng(1).permutation(xg.shape[0])]
    lib = load_library()
    dev = torch.device("cuda", 0)
    dom = plan = SlabDomain.from_global(xg, box, (1, 1, 1), rc_list, dev)
    pair = PairANNP(1, device=0, style=style)
    pair.settings([])
    pair.coeff(["*", "*", pot, el])
    pair.init_style()
    h = pair.handle
    st = torch.cuda.current_stream(dev).cuda_stream
    pn, pf, pg, mx = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(0)
    assert lib.annp_hip_neigh_build_device(h, plan.nlocal, plan.nall, dom.x.data_ptr(), rc_list, C.byref(pn), C.byref(pf), C.byref(pg), C.byref(mx), st) == 0
    # rebuilds of the same list (the second and later ones take the pitched one-pass layout)
    tb = []
    for _ in range(4):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        assert lib.annp_hip_neigh_build_device(h, plan.nlocal, plan.nall, dom.x.data_ptr(), rc_list, C.byref(pn), C.byref(pf), C.byref(pg), C.byref(mx), st) == 0
        torch.cuda.synchronize(dev)
        tb.append((time.perf_counter() - t0) * 1e3)
    print("list builds (host clock, ms): first %.2f, rebuilds %s" % (tb[0], " ".join("%.2f" % v for v in tb[1:])))
    eng = torch.zeros(1, dtype=torch.float64, device=dev)
    vir = torch.zeros(6, dtype=torch.float64, device=dev) if os.environ.get("KBENCH_VIRIAL") else None
    warm = 3
    for k in range(reps + warm):
        if k == warm:
            # (the first evaluations size the state the later ones run with -- the flag words come back without anybody waiting, so the
            # second may still run with the first one's room: the timed ones start once the handle has seen them)
            assert lib.annp_hip_sync(h) == 0, lib.annp_hip_last_error(h)
            lib.annp_hip_set_timing(h, 1)
        dom.f.zero_()
        eng.zero_()
        rc = lib.annp_hip_compute_device(h, plan.nlocal, plan.nall, dom.x.data_ptr(), None, None, pn, pf, pg, mx.value, dom.f.data_ptr(), None, eng.data_ptr(), vir.data_ptr() if vir is not None else None, None, st)
        assert rc == 0, lib.annp_hip_last_error(h)
    ms = np.zeros(4)
    ns = C.c_int(0)
    lib.annp_hip_timing_stats(h, ms.ctypes.data_as(C.POINTER(C.c_double)), C.byref(ns))
    print("%s atoms=%d ghosts=%d maxnbr=%d  desc %.3f  net %.3f  force %.3f  total %.3f ms  -> %.2f M atom-evals/s  E/atom %.6f" % (
        kind, plan.nlocal, plan.nghost, mx.value, ms[0], ms[1], ms[2], ms[3], plan.nlocal / ms[3] / 1e3, float(eng.item()) / plan.nlocal))
    print("eval_path %d" % lib.annp_hip_eval_path(h))


if __name__ == "__main__":
    main()
