"""A C++ host stepping MD on the device through include/annp_hip.h alone -- tests/cpp/annp_md_driver.cpp, no Python and
no torch in that process: annp_hip_verlet_half, _halo_unpack_images, _neigh_build_device, _compute_device, _reverse_fold in
Verlet::run's order, a reneighbouring every few steps.  Checked here: the first E_pair against the oracle at the same
positions, and E_pair + E_kin over the run (forces are the gradient of the energy the library reports, so the total only
shows velocity-Verlet's O(dt^2) error: four times smaller at half the step)."""
import os
import subprocess

import numpy as np
import pytest

from annp_testlib import A_FE, FAST, FE_POT, KIND_FE, ROOT, System, bcc, oracle_compute, perturb

DRIVER = os.path.join(ROOT, "tests", "cpp", "annp_md_driver")


def build_driver():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpp"), "annp_md_driver"])
    return DRIVER


def test_md_driver_uses_only_the_c_abi_and_the_hip_runtime():
    """what the driver leaves undefined: annp_* entries, the PairANNP mirror, hip* runtime calls, libc/libstdc++ -- nothing else
    of this repository, and no Python"""
    build_driver()
    und = subprocess.check_output(["nm", "-uC", DRIVER], text=True)
    mine = [ln.split(" U ", 1)[1].strip() for ln in und.splitlines() if "annp" in ln]
    assert any(s.startswith("annp_hip_compute_device") for s in mine) and any(s.startswith("annp_hip_verlet_half") for s in mine)
    assert all(s.startswith("annp_hip_") or s.startswith("annp_host::PairANNP::") for s in mine), mine
    needed = subprocess.check_output(["readelf", "-d", DRIVER], text=True)
    assert "libannp_hip.so" in needed and "libamdhip64" in needed and "python" not in needed and "torch" not in needed


def run(tmp_path, cells, steps, dt, every, tag):
    out = tmp_path / ("md_%s.txt" % tag)
    subprocess.run([build_driver(), FE_POT, "Fe", str(cells), str(steps), repr(dt), str(every), str(out)], check=True, timeout=600)
    return np.loadtxt(out)


@pytest.mark.gpu
def test_cpp_host_md_energy_and_conservation(tmp_path, fe_pot):
    cells = 8
    a = run(tmp_path, cells, 40, 0.001, 5, "a")
    b = run(tmp_path, cells, 80, 0.0005, 10, "b")
    c = run(tmp_path, cells, 40, 0.001, 0, "c")                 # never re-planned: the list of step 0 (skin 2 A) all the way
    x0, box = bcc(cells, cells, cells, A_FE)
    o = oracle_compute(fe_pot, System(perturb(x0, 12345, 0.05), box), KIND_FE, FAST)
    assert abs(a[0, 1] - o["energy"]) < 1e-6                    # BASELINE.json's energy tolerance, on the total of 1024 atoms
    assert abs(a[0, 1] - o["energy"]) < 1e-11 * abs(o["energy"])
    # run to run: the order of the energy atomics moves the last bit of a 4.6e6 eV total (atomic reference energies included)
    assert max(abs(a[0, 1] - b[0, 1]), abs(a[0, 1] - c[0, 1])) < 1e-14 * abs(a[0, 1]) and a[0, 2] == 0.0
    ea, eb = a[:, 1] + a[:, 2], b[:, 1] + b[:, 2]
    ke = a[:, 2].max()
    assert ke > 5.0                                              # eV moved into kinetic energy by the relaxing lattice
    da, db = np.abs(ea - ea[0]).max(), np.abs(eb - eb[0]).max()
    assert da < 2e-3 * ke and db < 0.35 * da
    # re-planning (wrap, new images, new list) changes nothing but the order of sums
    assert np.abs(a[:, 1] - c[:, 1]).max() < 1e-9 * abs(a[0, 1]) and np.abs(a[:, 2] - c[:, 2]).max() < 1e-9
    assert a[:, 3].min() > 1024                                  # images: more ghosts than owned atoms in a box this small
