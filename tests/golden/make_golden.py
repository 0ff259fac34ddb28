#!/usr/bin/env python3
"""Regenerates tests/golden/annp_golden.npz from the CPU oracle (LITERAL strategy: one atom at a
time, dG materialised, operations in the reference's order).  The oracle itself is pinned to the
reference by tests/test_oracle_pins.py; these vectors freeze its output so that the HIP path (and
the oracle's FAST strategy) are checked against committed numbers, not only against a freshly
built oracle.  Inputs are regenerated from seeds by the tests, outputs are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from annp_testlib import (A_FE, A_NI, ANNA_POT, FE_POT, KIND_FE, KIND_NI_COMPAT, KIND_NI_FIXED, LITERAL, NI_POT, System, anna_compute,  # noqa: E402
                          bcc, fcc, oracle_compute, perturb, read_anna, read_pot)

CASES = {
    # name: (lattice, cells, a, seed, amplitude, potential, kind)
    "fe_4x4x4": ("bcc", (4, 4, 4), A_FE, 12345, 0.05, FE_POT, KIND_FE),
    "fe_3x4x5_big_disp": ("bcc", (3, 4, 5), A_FE, 777, 0.15, FE_POT, KIND_FE),
    "ni_3x3x3_fixed": ("fcc", (3, 3, 3), A_NI, 4242, 0.05, NI_POT, KIND_NI_FIXED),
    "ni_3x3x3_compat": ("fcc", (3, 3, 3), A_NI, 4242, 0.05, NI_POT, KIND_NI_COMPAT),
}


def build(case):
    lat, cells, a, seed, amp, potfile, kind = CASES[case]
    x, box = (bcc if lat == "bcc" else fcc)(*cells, a)
    return System(perturb(x, seed, amp), box), read_pot(potfile), kind


# pair_style anna_adp: (cells, seed, amplitude, list cutoff)
ANNA_CASES = {"anna_4x4x4": ((4, 4, 4), 2310, 0.06, 7.055), "anna_3x5x4_big_disp": ((3, 5, 4), 99, 0.2, 7.055)}


def build_anna(case):
    cells, seed, amp, rc_list = ANNA_CASES[case]
    x, box = bcc(*cells, A_FE)
    return System(perturb(x, seed, amp), box, rc_list=rc_list), read_anna(ANNA_POT)


def main():
    out = {}
    for name in ANNA_CASES:
        s, pot = build_anna(name)
        r = anna_compute(pot, s, want_virial=True)
        for key in ("eatom", "f", "f_all", "virial", "G", "lparams"):
            out[name + "/" + key] = r[key]
        print("%-22s nlocal %4d nall %5d  E %.9f  |F|max %.6f" % (name, s.nlocal, s.nall, r["energy"], np.abs(r["f"]).max()))
    np.savez_compressed(os.path.join(HERE, "anna_golden.npz"), **out)
    out = {}
    for name in CASES:
        s, pot, kind = build(name)
        r = oracle_compute(pot, s, kind, LITERAL, want_virial=True, want_G=True)
        out[name + "/eatom"] = r["eatom"]
        out[name + "/f"] = r["f"]                 # ghost forces folded onto owners
        out[name + "/f_all"] = r["f_all"]         # as Pair::compute leaves atom->f (ghosts included)
        out[name + "/virial"] = r["virial"]
        out[name + "/G"] = r["G"]
        out[name + "/dEdG"] = r["dEdG"]
        print("%-22s nlocal %4d nall %5d  E %.9f  |F|max %.6f" % (name, s.nlocal, s.nall, r["energy"], np.abs(r["f"]).max()))
    np.savez_compressed(os.path.join(HERE, "annp_golden.npz"), **out)


if __name__ == "__main__":
    main()
