"""pair_style anna_adp: the CPU oracle's self-consistency.

The reference ships no test, log or golden value for this pair style and cannot be built here, so its oracle is
"parity unpinned" upstream (oracle/anna_oracle.h).  What can be checked without the reference: the file is read
as the reference's parser would read it, and the restated force loop (adp:215-280) is what it claims to be --
the exact gradient of the restated energy (adp:165-212) with the two network outputs held fixed."""
import numpy as np
import pytest

from annp_testlib import A_FE, ANNA_POT, System, anna_compute, bcc, perturb, read_anna


@pytest.fixture(scope="module")
def pot():
    return read_anna(ANNA_POT)


def test_file_values(pot):
    assert (pot.ntl, pot.nhl, pot.nnod, pot.nout, pot.nsf, pot.npsf, pot.ntsf, pot.ngp) == (4, 2, 6, 2, 28, 9, 19, 17)
    assert list(pot.flagact)[:3] == [4, 4, 0]
    assert pot.cut == 5.055 and pot.e_base == -4473.0075 and pot.e_scal == 1.0 and pot.mass == 55.847
    g = list(pot.gparams)[:17]
    assert g[0] == -9.46e-04 and g[6] == -3.2461 and g[10] == -4.6719 and g[12] == 1.65 and g[16] == 0.1086
    assert pot.W[0][0] == -0.405974984 and pot.W[2][6] == -0.83049798        # last layer: 2 rows of 6
    assert pot.B[2][0] == 0.039707493 and pot.B[2][1] == 0.496466488


def test_descriptor_is_the_raw_chebyshev_sum(pot):
    """perfect bcc, a = 2.8553: 58 neighbours inside 5.055 A; G_0 = sum fc, G_9 = sum_{j<k} fc fc (T_0 = 1)"""
    x0, box = bcc(4, 4, 4, A_FE)
    s = System(x0, box, rc_list=7.0)
    o = anna_compute(pot, s, inum=1)
    d = s.x[s.neigh[s.first[0]: s.first[0] + s.numneigh[0]]] - s.x[0]
    r = np.sqrt((d * d).sum(1))
    r = r[r <= pot.cut]
    assert len(r) == 58
    fc = 0.5 * (np.cos(np.pi * r / pot.cut) + 1.0)
    assert abs(o["G"][0, 0] - fc.sum()) < 1e-12
    assert abs(o["G"][0, 9] - 0.5 * (fc.sum() ** 2 - (fc * fc).sum())) < 1e-10


def test_forces_are_the_frozen_parameter_gradient(pot):
    x0, box = bcc(3, 3, 3, A_FE)
    xp = perturb(x0, 9, 0.1)
    s = System(xp, box, rc_list=6.5)
    frozen = [0.9, 1.3]
    o = anna_compute(pot, s, frozen=frozen)
    assert np.abs(o["f"].sum(0)).max() < 1e-10                               # pairwise equal and opposite
    h = 1e-5
    for (a, c) in [(0, 0), (7, 1), (20, 2), (41, 0)]:
        e = []
        for sgn in (+1, -1):
            xq = xp.copy()
            xq[a, c] += sgn * h
            s.refresh_ghosts(xq)
            e.append(float((anna_compute(pot, s, frozen=frozen)["eatom"] - pot.e_base).sum()))   # (the base would drown the difference)
        s.refresh_ghosts(xp)
        fd = -(e[0] - e[1]) / (2 * h)
        assert abs(fd - o["f"][a, c]) < 2e-6 * max(1.0, abs(fd)), (a, c, fd, o["f"][a, c])


def test_translation_and_virial_bookkeeping(pot):
    x0, box = bcc(3, 3, 3, A_FE)
    xp = perturb(x0, 21, 0.07)
    s = System(xp, box, rc_list=6.5)
    o = anna_compute(pot, s, want_virial=True, want_vatom=True)
    s2 = System(xp + np.array([0.123, -0.4, 0.77]), box + np.array([0.123, -0.4, 0.77, 0.123, -0.4, 0.77]), rc_list=6.5)
    o2 = anna_compute(pot, s2)
    assert abs(o["energy"] - o2["energy"]) < 1e-8 and np.abs(o["f"] - o2["f"]).max() < 1e-9
    assert np.abs(o["vatom"].sum(0) - o["virial"]).max() < 1e-9              # per-atom shares add up to the tally
    assert np.all(np.isfinite(o["lparams"])) and o["lparams"].shape == (s.nlocal, 2)
