"""Synthetic inputs of the benchmark and of the parity tests: perfect bcc / fcc lattices in cell-major order (what LAMMPS'
create_atoms + atom_modify sort leave: atoms that are close in space are close in index) with counter-based displacements
(element-wise splitmix64, so that any slice of a box can be generated without the rest), and where the
potential files live.  No arithmetic of the hot path in here."""
import gzip
import os

import numpy as np

DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")       # the reference's own data files (MPL-2.0 notice beside them)
POTENTIALS = os.path.join(DATA, "potentials")
FE_ST = os.path.join(DATA, "fe_st.dat.gz")        # the reference's published benchmark configuration (perf zip fe_st.dat, gzip)
FE_POT = os.path.join(POTENTIALS, "fe_annp_potential_2.ann")
NI_POT = os.path.join(POTENTIALS, "ni_annp_potential_2.ann")
ANNA_POT = os.path.join(POTENTIALS, "fe_adp_potential_2310.anna")

A_FE = 2.8553   # bcc Fe lattice constant used by the reference's own generator
A_NI = 3.52


def splitmix64(z):
    """Counter-based generator: element-wise splitmix64 finaliser."""
    z = (z + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def uniform_counter(n, seed):
    """n doubles in [0,1) from counters 0..n-1."""
    with np.errstate(over="ignore"):
        z = splitmix64(np.arange(n, dtype=np.uint64) ^ np.uint64(seed))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def bcc(nx, ny, nz, a):
    """bcc cells in cell-major order (both basis atoms of a cell adjacent)."""
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    cells = np.stack([i.ravel(), j.ravel(), k.ravel()], axis=1).astype(np.float64)
    x = np.empty((cells.shape[0], 2, 3))
    x[:, 0, :] = cells
    x[:, 1, :] = cells + 0.5
    box = np.array([0, 0, 0, nx * a, ny * a, nz * a], dtype=np.float64)
    return (x.reshape(-1, 3) * a).copy(), box


def fcc(nx, ny, nz, a):
    i, j, k = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    cells = np.stack([i.ravel(), j.ravel(), k.ravel()], axis=1).astype(np.float64)
    basis = np.array([[0, 0, 0], [0.5, 0.5, 0], [0.5, 0, 0.5], [0, 0.5, 0.5]])
    x = cells[:, None, :] + basis[None, :, :]
    box = np.array([0, 0, 0, nx * a, ny * a, nz * a], dtype=np.float64)
    return (x.reshape(-1, 3) * a).copy(), box


def perturb(x, seed=12345, amp=0.05):
    u = uniform_counter(x.size, seed).reshape(x.shape)
    return x + (2.0 * u - 1.0) * amp


def lammps_sort_order(x, box, binsize=4.25, seed=7):
    """The order `atom_modify sort` (LAMMPS' default: every 1000 steps, bins of half the neighbour cutoff) leaves atoms in: by spatial
    bin, x fastest (Atom::sort: ibin = iz * nbiny * nbinx + iy * nbinx + ix), in no particular order inside a bin (here: random).
    Returns the permutation; x[perm] is the sorted array."""
    lo, hi = np.asarray(box[:3]), np.asarray(box[3:])
    nb = np.maximum(1, np.floor((hi - lo) / binsize).astype(np.int64))
    ib = np.minimum(nb - 1, np.maximum(0, np.floor((x - lo) / (hi - lo) * nb).astype(np.int64)))
    key = (ib[:, 2] * nb[1] + ib[:, 1]) * nb[0] + ib[:, 0]
    tie = np.random.default_rng(seed).permutation(x.shape[0])
    return np.lexsort((tie, key))


def load_fe_st():
    """The reference's own benchmark configuration (perf zip fe_st.dat, 152 880 Fe atoms, `boundary m p m`): positions in id
    order, box (xlo, ylo, zlo, xhi, yhi, zhi)."""
    with gzip.open(FE_ST, "rt") as fh:
        lines = fh.read().split("\n")
    n = int(lines[1].split()[0])
    xlo, xhi = map(float, lines[3].split()[:2])
    ylo, yhi = map(float, lines[4].split()[:2])
    zlo, zhi = map(float, lines[5].split()[:2])
    start = next(i for i, l in enumerate(lines) if l.startswith("Atoms")) + 2
    arr = np.loadtxt(lines[start:start + n])
    x = np.ascontiguousarray(arr[np.argsort(arr[:, 0]), 2:5])
    return x, np.array([xlo, ylo, zlo, xhi, yhi, zhi])
