"""The Chebyshev passes without a pair loop (meng_zhang_amd/csrc/fe_sh_kernels.hpp: moments of the neighbourhood, Legendre
addition theorem; descriptor pass annp_fe_desc_sh, force pass annp_fe_force_sh) against the pair-loop kernels they replace
(ANNP_HIP_FE_DESC=pairs ANNP_HIP_FE_FORCE=pairs) and against the literal oracle (fe_v2/src/pair_annp.cpp:633-695 and 190-213
restated, oracle/annp_oracle.c): rows of raw sums through annp_hip_last_descriptors, then energies and forces.  Also the atoms
the moment kernels hand to the fix-up launches."""
import ctypes as C
import os

import numpy as np
import pytest

from annp_testlib import A_FE, FAST, FE_POT, KIND_FE, LITERAL, System, bcc, oracle_compute, perturb

pytestmark = pytest.mark.gpu


def make_pair(**env):
    from meng_zhang_amd import PairANNP
    old = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        p = PairANNP(ntypes=1, device=0)
        p.settings([])
        p.coeff(["*", "*", FE_POT, "Fe"])
        p.init_style()          # the switches are read here
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return p


def evaluate(pair, s):
    from meng_zhang_amd import AtomData, NeighList
    from meng_zhang_amd.lib import load_library
    pair.atom = AtomData(s.x, s.nlocal, s.type)
    pair.list = NeighList(s.ilist, s.numneigh, s.first, s.neigh)
    pair.ago = 0
    e = pair.compute(eflag=1, vflag=0, eflag_atom=True)
    rows = np.zeros((s.inum, 32))
    lib = load_library()
    assert lib.annp_hip_last_descriptors(pair.handle, rows.ctypes.data_as(C.POINTER(C.c_double)), s.inum) == 0
    info = (C.c_int * 4)()
    assert lib.annp_hip_eval_info(pair.handle, info) == 0
    return dict(e=e, f=s.fold(pair.atom.f), eatom=pair.eatom[: s.nlocal].copy(), rows=rows, nmax=info[0])


def raw_rows_of_oracle(pot, o):
    """the oracle's descriptors are normalised (fe:98-108, 178-180): G = (raw - norm1) * scale"""
    n0, n1 = np.array(pot.norm0[: pot.nsf]), np.array(pot.norm1[: pot.nsf])
    t = np.sqrt(n0 - n1 * n1)
    assert (t > 1e-10).all()
    return o["G"] * t + n1


@pytest.mark.parametrize("shape,amp", [((6, 6, 6), 0.05), ((5, 4, 3), 0.3), ((7, 3, 3), 0.0)])
def test_rows_equal_pair_loop_and_oracle(fe_pot, shape, amp):
    x0, box = bcc(*shape, A_FE)
    s = System(perturb(x0, 3, amp) if amp else x0, box)
    a, b = make_pair(), make_pair(ANNP_HIP_FE_DESC="pairs", ANNP_HIP_FE_FORCE="pairs")
    try:
        ra, rb = evaluate(a, s), evaluate(b, s)
    finally:
        a.close()
        b.close()
    # a column's own size, but not less than 1: in a perfect lattice the high orders cancel to 1e-4 out of terms of 1e3
    scale = np.maximum(np.abs(rb["rows"]).max(axis=0), 1.0)
    assert (ra["rows"][:, 28:] == 0.0).all()
    # same sums, different order and a different (exact) algebra: a few 1e-13 of the column's size
    assert (np.abs(ra["rows"] - rb["rows"]) / scale).max() < 2e-12
    o = oracle_compute(fe_pot, s, KIND_FE, LITERAL, want_G=True)
    raw = raw_rows_of_oracle(fe_pot, o)
    assert (np.abs(ra["rows"][:, :28] - raw) / scale[:28]).max() < 2e-12
    assert abs(ra["e"] - o["energy"]) < 1e-6 and np.abs(ra["eatom"] - o["eatom"]).max() < 1e-6
    assert np.abs(ra["f"] - o["f"]).max() < 1e-5
    assert np.abs(ra["f"] - rb["f"]).max() < 1e-9 * max(1.0, np.abs(rb["f"]).max())


@pytest.mark.parametrize("cap", [16, 64, 96])
def test_atoms_beyond_the_state_area_go_through_the_fixup_launch(fe_pot, cap):
    """ANNP_HIP_SH_CAP = state slots per atom of the first evaluation.  A box whose atoms have ~112 neighbours with room
    for 16/64/96: every atom is queued and evaluated by the pair-loop kernel, same rows.  The second evaluation has adapted."""
    x0, box = bcc(5, 5, 5, A_FE)
    s = System(perturb(x0, 11, 0.1), box)
    a, b = make_pair(ANNP_HIP_SH_CAP=cap), make_pair()
    try:
        r1 = evaluate(a, s)
        a.eatom[:] = 0.0
        r2 = evaluate(a, s)
        rb = evaluate(b, s)
    finally:
        a.close()
        b.close()
    assert r1["nmax"] > 96
    scale = np.maximum(np.abs(rb["rows"]).max(axis=0), 1.0)
    for r in (r1, r2):
        assert (np.abs(r["rows"] - rb["rows"]) / scale).max() < 2e-12
        assert abs(r["e"] - rb["e"]) < 1e-9 * abs(rb["e"])
    o = oracle_compute(fe_pot, s, KIND_FE, FAST)
    assert abs(r1["e"] - o["energy"]) < 1e-6 and np.abs(r1["f"] - o["f"]).max() < 1e-5


@pytest.mark.parametrize("dmin,edge,nmin,nmax", [(1.5, 16.0, 129, 160), (1.38, 14.7, 161, 256)])
def test_mixed_queue_and_ragged_groups(fe_pot, dmin, edge, nmin, nmax):
    """a cluster with a dense core in vacuum, 4 does not divide the atom count.  Neighbour counts from a handful to ~145: above the 128
    slots of the force pass's two register turns, inside the 160 the moment kernels take since round 6 (the slots above 128 are an extra
    turn for two of a group's waves); and to ~185: core atoms exceed the state the moment kernels can be given and take the fix-up
    launch in the steady state too"""
    rng = np.random.default_rng(4)
    pts = []
    while len(pts) < 521:                       # random points, none closer than dmin
        c = rng.uniform(0, edge, 3)
        if all(np.sum((c - q) ** 2) > dmin ** 2 for q in pts):
            pts.append(c)
    x = np.array(pts) + 20.0
    box = np.array([0, 0, 0, 57.0, 57.0, 57.0])
    s = System(x, box, periodic=(0, 0, 0))
    a, b = make_pair(), make_pair(ANNP_HIP_FE_DESC="pairs", ANNP_HIP_FE_FORCE="pairs")
    try:
        ra = evaluate(a, s)
        a.eatom[:] = 0.0
        ra2 = evaluate(a, s)
        rb = evaluate(b, s)
    finally:
        a.close()
        b.close()
    assert nmin <= ra["nmax"] <= nmax and s.inum % 4 != 0
    scale = np.maximum(np.abs(rb["rows"]).max(axis=0), 1.0)
    for r in (ra, ra2):
        assert (np.abs(r["rows"] - rb["rows"]) / scale).max() < 2e-12
    o = oracle_compute(fe_pot, s, KIND_FE, FAST)
    for r in (ra, ra2):
        assert np.abs(r["eatom"] - o["eatom"]).max() < 1e-6 * max(1.0, np.abs(o["eatom"]).max())
        assert np.abs(r["f"] - o["f"]).max() < 1e-5 * max(1.0, np.abs(o["f"]).max())


def test_list_rows_longer_than_256_entries(fe_pot):
    """a 10.2 A list: ~400 entries per row, so the moment kernel reads the rows group by group instead of all four at once"""
    x0, box = bcc(8, 8, 8, A_FE)
    s = System(perturb(x0, 21, 0.08), box, rc_list=10.2)
    assert s.numneigh[: s.nlocal].max() > 256
    a, b = make_pair(), make_pair(ANNP_HIP_FE_DESC="pairs", ANNP_HIP_FE_FORCE="pairs")
    try:
        ra, rb = evaluate(a, s), evaluate(b, s)
    finally:
        a.close()
        b.close()
    scale = np.maximum(np.abs(rb["rows"]).max(axis=0), 1.0)
    assert (np.abs(ra["rows"] - rb["rows"]) / scale).max() < 2e-12
    o = oracle_compute(fe_pot, s, KIND_FE, FAST)
    assert abs(ra["e"] - o["energy"]) < 1e-6 and np.abs(ra["f"] - o["f"]).max() < 1e-9 * max(1.0, np.abs(o["f"]).max())


@pytest.mark.parametrize("env", [dict(ANNP_HIP_FE_FORCE="pairs"), dict(ANNP_HIP_FE_DESC="pairs")])
def test_one_pass_on_the_moments_the_other_on_the_pairs(fe_pot, env):
    """the switches are independent: moment descriptor pass + pair-loop force pass, and the pair-loop descriptor pass (which
    leaves no moments, so the force pass is the pair loop too)"""
    x0, box = bcc(6, 5, 4, A_FE)
    s = System(perturb(x0, 5, 0.15), box)
    a = make_pair(**env)
    try:
        r1 = evaluate(a, s)
        a.eatom[:] = 0.0
        r2 = evaluate(a, s)
    finally:
        a.close()
    o = oracle_compute(fe_pot, s, KIND_FE, FAST)
    for r in (r1, r2):
        assert abs(r["e"] - o["energy"]) < 1e-6 and np.abs(r["f"] - o["f"]).max() < 1e-9 * max(1.0, np.abs(o["f"]).max())


def test_rows_do_not_depend_on_the_orientation(fe_pot):
    """the moments single out the z axis (and x + iy), the sums they stand for do not: the same cluster in three orientations --
    as built (lattice directions along the axes: neighbours exactly on the poles and on the equator), rotated by a random
    matrix, and with x, y, z permuted -- gives the same rows and the same energy"""
    x0, _ = bcc(5, 5, 5, A_FE)
    x0 = perturb(x0, 9, 0.02) - x0.mean(axis=0)
    rng = np.random.default_rng(12)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    box = np.array([0, 0, 0, 60.0, 60.0, 60.0])
    rows, energies = [], []
    a = make_pair()
    try:
        for rot in (np.eye(3), q, np.eye(3)[[2, 0, 1]]):
            s = System(x0 @ rot.T + 30.0, box, periodic=(0, 0, 0))
            a.eatom = None
            r = evaluate(a, s)
            rows.append(r["rows"])
            energies.append(r["e"])
    finally:
        a.close()
    scale = np.maximum(np.abs(rows[0]).max(axis=0), 1.0)
    for other in rows[1:]:
        assert (np.abs(other - rows[0]) / scale).max() < 2e-12
    assert max(abs(e - energies[0]) for e in energies) < 1e-9 * abs(energies[0])


def test_atoms_without_neighbours(fe_pot):
    """isolated atoms (and a dimer, and a small cluster) on a fresh handle: an atom without neighbours has no row in the list the
    descriptor pass hands to the force pass, no moments worth the name, zero force and the energy of an empty environment"""
    rng = np.random.default_rng(3)
    cluster = rng.uniform(0, 6.0, (23, 3))
    x = np.vstack([cluster + 40.0, [[10.0, 10.0, 10.0]], [[70.0, 12.0, 33.0]], [[20.0, 60.0, 60.0], [22.4, 60.0, 60.0]]])
    s = System(x, np.array([0, 0, 0, 90.0, 90.0, 90.0]), periodic=(0, 0, 0))
    assert s.numneigh[: s.nlocal].min() == 0
    a = make_pair()
    try:
        r = evaluate(a, s)
    finally:
        a.close()
    o = oracle_compute(fe_pot, s, KIND_FE, FAST)
    assert np.abs(r["eatom"] - o["eatom"]).max() < 1e-6 * max(1.0, np.abs(o["eatom"]).max())
    assert np.abs(r["f"] - o["f"]).max() < 1e-9 * max(1.0, np.abs(o["f"]).max())
    assert np.abs(r["f"][23:25]).max() == 0.0 and (r["rows"][23:25] == 0.0).all()


@pytest.mark.parametrize("ragged", [False, True])
def test_grouped_change_of_basis_gives_the_parked_rows(fe_pot, ragged):
    """ANNP_HIP_SH_TAIL=group (round 6, a developer switch: measured 2 % slower than the default): launches with room for at most 112
    neighbours change basis in three groups of columns out of LDS instead of parking their totals in the moment row.  Same descriptor
    rows, energies and forces -- the second evaluation is the one that runs it (the first has room for 128) -- on a box and on a
    ragged cluster whose atom count 4 does not divide (rows of atoms that do not exist, groups with a single neighbour)."""
    if ragged:
        rng = np.random.default_rng(9)
        pts = []
        while len(pts) < 203:
            c = rng.uniform(0, 15.0, 3)
            if all(np.sum((c - q) ** 2) > 2.0 ** 2 for q in pts):
                pts.append(c)
        s = System(np.array(pts) + 20.0, np.array([0, 0, 0, 55.0, 55.0, 55.0]), periodic=(0, 0, 0))
    else:
        x0, box = bcc(6, 5, 4, A_FE)
        s = System(perturb(x0, 21, 0.1), box)
    out = {}
    for tail in ("group", "park"):
        p = make_pair(ANNP_HIP_SH_TAIL=tail)
        try:
            evaluate(p, s)
            p.eatom[:] = 0.0
            out[tail] = evaluate(p, s)
        finally:
            p.close()
    assert out["group"]["nmax"] <= 112
    scale = np.maximum(np.abs(out["park"]["rows"]).max(axis=0), 1.0)
    assert (np.abs(out["group"]["rows"] - out["park"]["rows"]) / scale).max() < 2e-12
    o = oracle_compute(fe_pot, s, KIND_FE, FAST)
    for r in out.values():
        assert np.abs(r["eatom"] - o["eatom"]).max() < 1e-6 * max(1.0, np.abs(o["eatom"]).max())
        assert np.abs(r["f"] - o["f"]).max() < 1e-9 * max(1.0, np.abs(o["f"]).max())
