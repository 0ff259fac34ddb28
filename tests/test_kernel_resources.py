"""What hipcc reports for the gfx950 kernels (`make -C meng_zhang_amd/csrc asm`, -Rpass-analysis=kernel-resource-usage; no GPU
needed): nothing spills to scratch, and the occupancies DESIGN.md quotes are the compiler's.  Round 2 shipped
annp_ni_force with 24 VGPRs spilled and annp_anna_adp with 132 B of private arrays while DESIGN said "no scratch"."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.fixture(scope="module")
def kernels():
    import kernel_resources
    return {k["kernel"]: k for k in kernel_resources.collect()}


def test_no_kernel_uses_scratch(kernels):
    assert len(kernels) >= 40
    bad = {n: (k["scratch"], k["vgpr_spill"]) for n, k in kernels.items() if k["scratch"] or k["vgpr_spill"]}
    assert not bad, bad


def test_occupancies_quoted_in_design(kernels):
    def occ(prefix):
        hits = [k for n, k in kernels.items() if n.startswith(prefix)]
        assert hits, prefix
        return {(k["vgpr"], k["occupancy"]) for k in hits}
    shipped = "3, 24, 2, 3, 4, 268698113u, 328193u"
    assert all(w == 4 and v <= 128 for v, w in occ("annp::annp_ni_desc<%s, false, " % shipped))      # record capacity compiled in (20) and at run time (0)
    assert all(w == 3 and v <= 168 for v, w in occ("annp::annp_ni_force<%s" % shipped))
    assert all(w >= 4 and v <= 128 for v, w in occ("annp::annp_fe_desc<9, 19>"))
    assert all(w == 3 and v <= 168 for v, w in occ("annp::annp_fe_desc_sh<9, 19>"))      # 13 KB of LDS per wave: 12 waves per CU either way
    assert all(w == 4 and v <= 128 for v, w in occ("annp::annp_fe_force_sh<9, 19"))      # two 8-wave workgroups per CU
    assert all(w >= 5 for v, w in occ("annp::annp_fe_force<9, 19, false, true, 128>"))
    assert all(w >= 4 for v, w in occ("annp::annp_mlp_mfma<7, 2, 3>"))      # 8 waves per workgroup: at least 2 workgroups of registers
    assert all(w == 3 for v, w in occ("annp::annp_anna_adp<"))
