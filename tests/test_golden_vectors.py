"""Committed golden vectors (tests/golden/annp_golden.npz, made by tests/golden/make_golden.py from the
oracle's LITERAL strategy): the oracle's FAST strategy on CPU and the HIP path on GPU must reproduce them."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from make_golden import ANNA_CASES, CASES, build, build_anna  # noqa: E402

from annp_testlib import FAST, GOLDEN, KIND_FE, KIND_NI_COMPAT, LITERAL, oracle_compute  # noqa: E402

GOLD = np.load(os.path.join(GOLDEN, "annp_golden.npz"))
GOLD_ANNA = np.load(os.path.join(GOLDEN, "anna_golden.npz"))


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("strategy", [LITERAL, FAST])
def test_oracle_reproduces_golden(case, strategy):
    s, pot, kind = build(case)
    r = oracle_compute(pot, s, kind, strategy, want_virial=True, want_G=True)
    tol = 0.0 if strategy == LITERAL else 1e-11
    for key in ("eatom", "f", "f_all", "virial", "G", "dEdG"):
        ref = GOLD[case + "/" + key]
        assert np.abs(r[key] - ref).max() <= tol * max(1.0, np.abs(ref).max()), key


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(CASES))
def test_hip_reproduces_golden(case):
    from meng_zhang_amd import AtomData, NeighList, PairANNP
    s, pot, kind = build(case)
    potfile = CASES[case][5]
    p = PairANNP(1, device=0)
    p.settings([])
    p.coeff(["*", "*", potfile, "Fe" if kind == KIND_FE else "Ni"])
    p.set_ni_compat(kind == KIND_NI_COMPAT)
    p.init_style()
    p.atom = AtomData(s.x, s.nlocal, s.type)
    p.list = NeighList(s.ilist, s.numneigh, s.first, s.neigh)
    p.compute(eflag=1, vflag=1, eflag_atom=True)
    try:
        assert np.abs(p.eatom[: s.nlocal] - GOLD[case + "/eatom"]).max() < 1e-6          # BASELINE: 1e-6 eV
        assert np.abs(p.atom.f - GOLD[case + "/f_all"]).max() < 1e-5                     # BASELINE: 1e-5 eV/A
        assert np.abs(p.atom.f - GOLD[case + "/f_all"]).max() < 1e-9                     # what fp64 actually gives
        assert np.abs(s.fold(p.atom.f) - GOLD[case + "/f"]).max() < 1e-9
        assert np.allclose(p.virial, GOLD[case + "/virial"], rtol=1e-9, atol=1e-9)
    finally:
        p.close()


@pytest.mark.parametrize("case", sorted(ANNA_CASES))
def test_anna_oracle_reproduces_golden(case):
    """pair_style anna_adp vectors (the oracle scatters forces from OpenMP threads: summation order, hence the last
    bits, may differ between runs)"""
    from annp_testlib import anna_compute
    s, pot = build_anna(case)
    r = anna_compute(pot, s, want_virial=True)
    for key in ("eatom", "G", "lparams"):
        assert np.array_equal(r[key], GOLD_ANNA[case + "/" + key]), key
    for key in ("f", "f_all", "virial"):
        ref = GOLD_ANNA[case + "/" + key]
        assert np.abs(r[key] - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max()), key


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(ANNA_CASES))
def test_anna_hip_reproduces_golden(case):
    from annp_testlib import ANNA_POT
    from meng_zhang_amd import AtomData, NeighList, PairANNP
    s, _ = build_anna(case)
    p = PairANNP(1, device=0, style="anna_adp")
    p.settings([])
    p.coeff(["*", "*", ANNA_POT, "Fe"])
    p.init_style()
    p.atom = AtomData(s.x, s.nlocal, s.type)
    p.list = NeighList(s.ilist, s.numneigh, s.first, s.neigh)
    p.compute(eflag=1, vflag=1, eflag_atom=True)
    try:
        assert np.abs(p.eatom[: s.nlocal] - GOLD_ANNA[case + "/eatom"]).max() < 1e-6
        assert np.abs(p.atom.f - GOLD_ANNA[case + "/f_all"]).max() < 1e-5
        assert np.abs(p.atom.f - GOLD_ANNA[case + "/f_all"]).max() < 1e-9 * max(1.0, np.abs(GOLD_ANNA[case + "/f_all"]).max())
        assert np.allclose(p.virial, GOLD_ANNA[case + "/virial"], rtol=1e-8, atol=1e-8)
    finally:
        p.close()
