#!/usr/bin/env python3
"""Developer tool: rate of the host-pointer entry point (annp_hip_compute: what LAMMPS calls), PCIe included."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from annp_testlib import A_FE, FE_POT, System, bcc, perturb
    from meng_zhang_amd import AtomData, NeighList, PairANNP
    cells = int(sys.argv[1]) if len(sys.argv) > 1 else 80
    x0, box = bcc(cells, cells, cells, A_FE)
    t = time.time()
    s = System(perturb(x0, 12345, 0.05), box)
    print("harness: nlocal %d nall %d list %.1f M entries, %.1f s" % (s.nlocal, s.nall, s.neigh.size / 1e6, time.time() - t))
    p = PairANNP(1, device=0)
    p.settings([])
    p.coeff(["*", "*", FE_POT, "Fe"])
    p.init_style()
    p.atom = AtomData(s.x, s.nlocal, s.type)
    t = time.time()
    p.list = NeighList(s.ilist, s.numneigh, s.first, s.neigh)
    print("firstneigh pointers: %.1f s" % (time.time() - t))
    p.compute(eflag=1, vflag=0, eflag_atom=False)            # first call: buffers, capacities
    for label, k, rebuild in (("ago=0 (list upload)", 3, True), ("ago>0", 5, False)):
        dt = 0.0
        for _ in range(k):
            p.atom.f[:] = 0.0               # (the caller's business, not timed)
            if rebuild:
                p.ago = 0
            t = time.perf_counter()
            e = p.compute(eflag=1, vflag=0, eflag_atom=False)
            dt += (time.perf_counter() - t) / k
        print("%-20s %.1f ms per call -> %.2f M atom-steps/s   E/atom %.6f" % (label, dt * 1e3, s.nlocal / dt / 1e6, e / s.nlocal))
    # device-built list (annp_hip_compute_n): only x and f cross PCIe
    p.ago = 0
    p.compute_n(cutneigh=8.5, eflag=1, vflag=0, eflag_atom=False)
    for label, k, rebuild in (("compute_n ago=0", 8, True), ("compute_n ago>0", 8, False)):
        dt, each = 0.0, []
        for _ in range(k):
            p.atom.f[:] = 0.0
            if rebuild:
                p.ago = 0
            t = time.perf_counter()
            e = p.compute_n(cutneigh=8.5, eflag=1, vflag=0, eflag_atom=False)
            each.append((time.perf_counter() - t) * 1e3)
            dt += each[-1] * 1e-3 / k
        print("%-20s %.1f ms per call -> %.2f M atom-steps/s   E/atom %.6f   (calls: %s)" % (
            label, dt * 1e3, s.nlocal / dt / 1e6, e / s.nlocal, " ".join("%.1f" % v for v in each)))
    # what an NPT step asks for (global virial) and what a step with per-atom energies asks for
    for label, kw in (("compute_n ago>0 vflag", dict(vflag=1, eflag_atom=False)), ("compute_n ago>0 eatom", dict(vflag=0, eflag_atom=True))):
        dt = 0.0
        for _ in range(4):
            p.atom.f[:] = 0.0
            if p.eatom is not None:
                p.eatom[:] = 0.0
            t = time.perf_counter()
            e = p.compute_n(cutneigh=8.5, eflag=1, **kw)
            dt += (time.perf_counter() - t) / 4
        print("%-22s %.1f ms per call -> %.2f M atom-steps/s   E/atom %.6f" % (label, dt * 1e3, s.nlocal / dt / 1e6, e / s.nlocal))
    if os.environ.get("ANNP_HIP_REGISTER") != "0":
        print("(caller's x and f page-locked in place; ANNP_HIP_REGISTER=0 for the staging route)")
    p.close()


if __name__ == "__main__":
    main()
