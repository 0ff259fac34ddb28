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
    assert all(w == 3 and v <= 168 for v, w in occ("annp::annp_fe_desc_sh<9, 19, "))      # 13 KB of LDS per wave: 12 waves per CU either way
    assert all(w == 4 and v <= 128 for v, w in occ("annp::annp_fe_force_sh<9, 19"))      # two 8-wave workgroups per CU
    assert all(w >= 5 for v, w in occ("annp::annp_fe_force<9, 19, false, true, 128>"))
    assert all(w >= 4 for v, w in occ("annp::annp_mlp_mfma<7, 2, 3>"))      # 8 waves per workgroup: at least 2 workgroups of registers
    assert all(w == 3 for v, w in occ("annp::annp_anna_adp<"))


# ---- the compiler workaround of the Chebyshev force pass, pinned ------------------------------------------------
# hipcc 7.2 builds `switch (place)` of annp_fe_force_sh wrongly when the place is a value it knows nothing about (DESIGN.md 4.3b
# "places"): one arm's moment loads go out from registers nobody has written.  `& 3` behind the register read makes it build the right
# decision tree -- a dependency on this compiler's behaviour.  tools/asm_check.py follows every definition that can reach an address
# register of a global-memory instruction through the kernel's control-flow graph, in the assembly `make asm` has just written.
CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "meng_zhang_amd", "csrc")


def test_force_pass_addresses_are_written_on_every_path(kernels):
    import asm_check
    res = asm_check.check(os.path.join(CSRC, "annp_hip.s"), ("annp_fe_force_sh<9, 19",))      # (the fixture has rebuilt the file)
    assert len(res) == 2, list(res)                     # with and without the virial tally
    for name, r in res.items():
        assert r["memory_instructions"] > 100 and r["instructions"] > 2000, (name, r["instructions"], r["memory_instructions"])
        assert not r["findings"], (name, r["findings"][:4])


def test_the_check_sees_the_construct_round_5_faulted_on(tmp_path):
    """the source without the `& 3` (-DANNP_SHF_NO_PLACE_MASK, built here and nowhere else): the check must report the loads that
    are reachable with nothing written to their address registers -- or this compiler no longer miscompiles the construct"""
    import subprocess

    import asm_check
    out = str(tmp_path / "nomask.s")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-fPIC", "-w",
           "-DANNP_SHF_NO_PLACE_MASK", "-S", "--cuda-device-only", os.path.join(CSRC, "annp_hip.hip"), "-o", out]
    subprocess.run(cmd, check=True, capture_output=True)
    res = asm_check.check(out, ("annp_fe_force_sh<9, 19",))
    found = [f for r in res.values() for f in r["findings"] if "nothing written" in f[3] and f[1].startswith("global_load")]
    if not found:
        pytest.skip("this hipcc builds the unmasked switch correctly: the `& 3` of fe_shf_kernels.hpp may go")
    assert len(found) >= 8          # eight loads of the light wave's arm, both halves of each address pair
