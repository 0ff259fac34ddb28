"""Network shapes, activation names and Chebyshev basis sizes the shipped potentials do not use
(SURVEY.md 8f.3): synthetic .ann files in the reference's layout, HIP path against the oracle."""
import numpy as np
import pytest

from annp_testlib import (A_FE, FAST, KIND_FE, LITERAL, System, bcc, oracle_compute, perturb, read_pot, write_ann)
from test_gpu_parity import make_pair, run

# Activation names: the reference probes every two-character window of the line (fe_v2/src/pair_annp.cpp:413-424),
# so "hyperbolic" would register hy AND li, "sigmoid" si AND mo.  The cases use the bare probes.
#   li -> 0 linear, hy -> 1 tanh, si -> 2 1/(1+exp(+a)), mo -> 3 1.7159 tanh(2a/3), ta -> 4 (... + 0.1 a)
CASES = {
    # name: (npsf, ntsf, nnod, ntl, activation names, expected flags)
    "hyper_sigmoid": (9, 19, 10, 4, ("hy", "si", "li"), [1, 2, 0]),
    "modified_tanh": (9, 19, 10, 4, ("mo", "ta", "li"), [3, 4, 0]),
    "one_hidden": (9, 19, 12, 3, ("ta", "li"), [4, 0]),
    "three_hidden_wide": (9, 19, 20, 5, ("mo", "hy", "si", "li"), [3, 1, 2, 0]),
    "small_basis": (6, 11, 7, 4, ("ta", "ta", "li"), [4, 4, 0]),
    "tiny_basis_wide_net": (3, 4, 32, 4, ("hy", "ta", "li"), [1, 4, 0]),
    "nonlinear_output": (9, 19, 10, 4, ("ta", "ta", "hy"), [4, 4, 1]),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_synthetic_potential_matches_oracle(name, tmp_path):
    npsf, ntsf, nnod, ntl, acts, flags = CASES[name]
    path = write_ann(str(tmp_path / (name + ".ann")), npsf, ntsf, nnod, ntl, acts, seed=len(name))
    pot = read_pot(path)
    assert (pot.npsf, pot.ntsf, pot.nnod, pot.ntl) == (npsf, ntsf, nnod, ntl)
    assert list(pot.flagact)[: ntl - 1] == flags
    x0, box = bcc(4, 4, 4, A_FE)
    s = System(perturb(x0, 77, 0.08), box, rc_list=8.5)
    o = oracle_compute(pot, s, KIND_FE, FAST, want_virial=True)
    p = make_pair(path, "Fe")
    try:
        r = run(p, s, vflag=1)
    finally:
        p.close()
    scale = max(1.0, np.abs(o["f"]).max())
    assert np.abs(r["eatom"] - o["eatom"]).max() < 1e-6 * max(1.0, np.abs(o["eatom"]).max() / 4000.0)
    assert np.abs(r["f"] - o["f"]).max() < 1e-5 * scale
    assert abs(r["energy"] - o["energy"]) < 1e-6 * s.nlocal
    assert np.abs(r["virial"] - o["virial"]).max() < 1e-5 * max(1.0, np.abs(o["virial"]).max())


@pytest.mark.gpu
def test_literal_strategy_agrees_on_a_synthetic_shape(tmp_path):
    """The oracle's FAST strategy is only trusted through LITERAL (the reference's own loop order): pin one
    synthetic shape against it as well."""
    path = write_ann(str(tmp_path / "lit.ann"), 6, 11, 7, 4, ("hy", "si", "li"), seed=5)
    pot = read_pot(path)
    x0, box = bcc(3, 3, 3, A_FE)
    s = System(perturb(x0, 3, 0.05), box, rc_list=8.5)
    a = oracle_compute(pot, s, KIND_FE, LITERAL)
    b = oracle_compute(pot, s, KIND_FE, FAST)
    assert np.abs(a["f"] - b["f"]).max() < 1e-9 and np.abs(a["eatom"] - b["eatom"]).max() < 1e-9
    p = make_pair(path, "Fe")
    try:
        r = run(p, s)
    finally:
        p.close()
    assert np.abs(r["eatom"] - a["eatom"]).max() < 1e-6
    assert np.abs(r["f"] - a["f"]).max() < 1e-5 * max(1.0, np.abs(a["f"]).max())


@pytest.mark.gpu
def test_oversized_shapes_are_refused(tmp_path):
    for kw in (dict(npsf=10, ntsf=19), dict(npsf=9, ntsf=20), dict(nnod=33)):
        args = dict(npsf=9, ntsf=19, nnod=10, ntl=4, acts=("ta", "ta", "li"))
        args.update(kw)
        path = write_ann(str(tmp_path / "big.ann"), **args)
        with pytest.raises(RuntimeError, match="code -9"):    # ANNP_HIP_ESHAPE, from annp_hip_init through init_style
            p = make_pair(path, "Fe")
            p.close()
