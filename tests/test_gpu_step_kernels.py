"""The MD step around the evaluation as library kernels (include/annp_hip.h: annp_hip_halo_pack, _halo_unpack_images,
_reverse_fold, _verlet_half -- what LAMMPS' Comm::forward_comm / reverse_comm and FixNVE do around Pair::compute for the
reference): each entry against a numpy restatement bit for bit, then SlabDomain driven by them against SlabDomain driven
by torch ops, on one rank and on virtual ranks, through a re-planning."""
import ctypes as C

import numpy as np
import pytest

from annp_testlib import A_FE, FAST, FE_POT, KIND_FE, System, ThreadFabric, bcc, oracle_compute, perturb, uniform_counter

pytestmark = pytest.mark.gpu
RC_LIST = 8.5


@pytest.fixture(scope="module")
def ctx():
    import torch
    from meng_zhang_amd import PairANNP
    from meng_zhang_amd.lib import load_library
    lib = load_library()
    pair = PairANNP(1, device=0)
    pair.settings([])
    pair.coeff(["*", "*", FE_POT, "Fe"])
    pair.init_style()
    yield torch, lib, pair.handle, torch.device("cuda", 0)
    pair.close()


def test_entries_match_numpy_bit_for_bit(ctx):
    torch, lib, h, dev = ctx
    rng = np.random.default_rng(5)
    n, nsend, nimg = 5000, 1777, 2311
    x = rng.normal(0, 10, (n + nimg, 3))
    st = torch.cuda.current_stream(dev).cuda_stream

    def T(a, dt=None):
        return torch.from_numpy(np.ascontiguousarray(a)).to(dev) if dt is None else torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(dt)

    # halo_pack
    idx = rng.integers(0, n, nsend).astype(np.int32)
    shift = rng.choice([-37.5, 0.0, 41.25], (nsend, 3))
    dx, out = T(x), torch.zeros((nsend, 3), dtype=torch.float64, device=dev)
    assert lib.annp_hip_halo_pack(h, nsend, T(idx).data_ptr(), T(shift).data_ptr(), dx.data_ptr(), out.data_ptr(), st) == 0
    assert np.array_equal(out.cpu().numpy(), x[idx] + shift)

    # halo_unpack_images + force clear
    root = rng.integers(0, n, nimg).astype(np.int32)
    ishift = rng.choice([-22.0, 0.0, 22.0], (nimg, 3))
    f = T(rng.normal(0, 1, (n + nimg, 3)))
    eng = T(np.array([3.25]))
    assert lib.annp_hip_halo_unpack_images(h, nimg, T(root).data_ptr(), T(ishift).data_ptr(), dx.data_ptr(), n, f.data_ptr(),
                                           3 * (n + nimg), eng.data_ptr(), st) == 0
    want = x.copy()
    want[n:] = x[root] + ishift
    assert np.array_equal(dx.cpu().numpy(), want)
    assert float(f.abs().max()) == 0.0 and float(eng.item()) == 0.0
    # ... clear only / images only
    f2 = T(np.ones((7, 3)))
    assert lib.annp_hip_halo_unpack_images(h, 0, None, None, None, 0, f2.data_ptr(), 21, None, st) == 0
    assert float(f2.abs().max()) == 0.0

    # reverse_fold: the order of np.add.at (ascending k per target)
    fa = rng.normal(0, 1, (n + nimg, 3))
    order = np.argsort(root, kind="stable")
    dst, cnt = np.unique(root[order], return_counts=True)
    start = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    df = T(fa)
    assert lib.annp_hip_reverse_fold(h, len(dst), T(dst.astype(np.int32)).data_ptr(), T(start).data_ptr(), T(order.astype(np.int32)).data_ptr(),
                                     df.data_ptr() + 24 * n, df.data_ptr(), st) == 0
    ref = fa.copy()
    np.add.at(ref, root, fa[n:])            # sequential, k ascending: the kernel's order
    assert np.array_equal(df.cpu().numpy(), ref)

    # verlet halves: product and sum rounded separately, as numpy does
    v = rng.normal(0, 1, (n, 3))
    ff = rng.normal(0, 1, (n, 3))
    xx = rng.normal(0, 10, (n, 3))
    dv, dff, dxx = T(v), T(ff), T(xx)
    dtf, dt = 0.8637e-2 * 0.5, 1e-3
    assert lib.annp_hip_verlet_half(h, n, dxx.data_ptr(), dv.data_ptr(), dff.data_ptr(), dtf, dt, st) == 0
    v1 = v + dtf * ff
    assert np.array_equal(dv.cpu().numpy(), v1) and np.array_equal(dxx.cpu().numpy(), xx + dt * v1)
    assert lib.annp_hip_verlet_half(h, n, None, dv.data_ptr(), dff.data_ptr(), dtf, 0.0, st) == 0
    assert np.array_equal(dv.cpu().numpy(), v1 + dtf * ff) and np.array_equal(dxx.cpu().numpy(), xx + dt * v1)
    # bad arguments are refused, nothing is launched
    assert lib.annp_hip_halo_pack(h, 5, None, None, None, None, st) == -1
    assert lib.annp_hip_reverse_fold(h, -1, None, None, None, None, None, st) == -1


@pytest.mark.parametrize("world", [1, 2, 4])
def test_domain_on_hip_kernels_equals_domain_on_torch_ops(ctx, fe_pot, world):
    """the same program twice -- SlabDomain with hip=(lib, handle) and with torch ops -- over a forward halo, an evaluation,
    the reverse halo, a Verlet step, a re-planning after atoms drifted across slab faces, and again"""
    torch, lib, _, dev = ctx
    from meng_zhang_amd import PairANNP
    from meng_zhang_amd.domain import SlabDomain
    x0, box = bcc(16, 5, 5, A_FE)
    xg = perturb(x0, 99, 0.05)
    drift = (2.0 * uniform_counter(xg.size, 17).reshape(xg.shape) - 1.0) * 0.9
    dtf, dt = 0.00432, 0.001

    def program(use_hip):
        def rank_program(rank, tp):
            pair = PairANNP(1, device=0)
            pair.settings([])
            pair.coeff(["*", "*", FE_POT, "Fe"])
            pair.init_style()
            h = pair.handle
            dom = SlabDomain.from_global(xg, box, (1, 1, 1), RC_LIST, dev, tp, extra={"v": np.zeros_like(xg)},
                                         hip=(lib, h) if use_hip else None)
            st = torch.cuda.current_stream(dev).cuda_stream
            eng = torch.ones(1, dtype=torch.float64, device=dev)
            pn, pf, pg, mx = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(0)
            snaps = []

            def evaluate():
                assert lib.annp_hip_compute_device(h, dom.nlocal, dom.nall, dom.x.data_ptr(), None, None, pn, pf, pg, mx.value,
                                                   dom.f.data_ptr(), None, eng.data_ptr(), None, None, st) == 0
                assert lib.annp_hip_sync(h) == 0

            for phase in range(2):
                if phase == 1:
                    dom.x[: dom.nlocal] += torch.from_numpy(drift[dom.ids.cpu().numpy()]).to(dev)
                    dom.replan()
                    eng.zero_()
                assert lib.annp_hip_neigh_build_device(h, dom.nlocal, dom.nall, dom.x.data_ptr(), RC_LIST, C.byref(pn), C.byref(pf),
                                                       C.byref(pg), C.byref(mx), st) == 0
                if phase == 0:
                    dom.forward(clear_forces=True, eng=eng)
                evaluate()
                dom.reverse()
                dom.verlet_half(dom.extra["v"], dtf, dt)
                dom.forward(clear_forces=True, eng=eng)
                x_after = dom.x.clone()
                assert float(dom.f.abs().max()) == 0.0 and float(eng.item()) == 0.0
                evaluate()
                dom.reverse()
                dom.verlet_half(dom.extra["v"], dtf, 0.0)
                torch.cuda.synchronize(dev)
                snaps.append(dict(ids=dom.ids.cpu().numpy(), x=x_after.cpu().numpy(), f=dom.f[: dom.nlocal].cpu().numpy(),
                                  v=dom.extra["v"].cpu().numpy(), e=float(eng.item()), nall=dom.nall))
            pair.close()
            return snaps
        return ThreadFabric(world).run(rank_program)

    a, b = program(True), program(False)
    for ra, rb in zip(a, b):
        for sa, sb in zip(ra, rb):
            assert np.array_equal(sa["ids"], sb["ids"]) and sa["nall"] == sb["nall"]
            # Positions of owned atoms, wire ghosts and images: gathers and single additions.  The integrator that produced
            # them read forces whose last bits depend on the order of the evaluation's atomics, in either engine.
            assert np.abs(sa["x"] - sb["x"]).max() < 1e-12
            scale = max(1.0, np.abs(sb["f"]).max())
            assert np.abs(sa["f"] - sb["f"]).max() < 1e-11 * scale
            assert np.abs(sa["v"] - sb["v"]).max() < 1e-12
            assert abs(sa["e"] - sb["e"]) < 1e-9 * abs(sb["e"])
    # and both are right: forces of the second phase against the single-domain oracle at those positions
    ids = np.concatenate([r[1]["ids"] for r in a])
    xs = np.concatenate([r[1]["x"][: len(r[1]["ids"])] for r in a])
    xfull = np.empty_like(xg)
    xfull[ids] = xs
    s = System(xfull, box)
    o = oracle_compute(fe_pot, s, KIND_FE, FAST)
    f = np.empty_like(xg)
    f[ids] = np.concatenate([r[1]["f"] for r in a])
    assert np.abs(f - o["f"]).max() < 1e-8 * max(1.0, np.abs(o["f"]).max())


def test_library_wire_sends_to_itself(ctx):
    """annp_hip_comm_*: the library's own RCCL caller (librccl opened at run time).  One rank is a whole communicator;
    a grouped send + receive with peer == own rank is what a slab that is its own periodic neighbour does."""
    torch, lib, _, dev = ctx
    from meng_zhang_amd import PairANNP
    pair = PairANNP(1, device=0)
    pair.settings([])
    pair.coeff(["*", "*", FE_POT, "Fe"])
    pair.init_style()
    h = pair.handle
    st = torch.cuda.current_stream(dev).cuda_stream
    a = torch.arange(3000, dtype=torch.float64, device=dev).reshape(1000, 3) * 0.25
    b = torch.zeros_like(a)
    kinds, cnts, peers = (C.c_int * 2)(1, 0), (C.c_longlong * 2)(3000, 3000), (C.c_int * 2)(0, 0)
    ptrs = (C.c_void_p * 2)(a.data_ptr(), b.data_ptr())
    assert lib.annp_hip_comm_route(h, 2, kinds, ptrs, cnts, peers, st) == -1            # no communicator yet
    assert b"annp_hip_comm_init" in lib.annp_hip_last_error(h)
    ident = C.create_string_buffer(128)
    assert lib.annp_hip_comm_unique_id(ident) == 0
    assert lib.annp_hip_comm_init(h, ident.raw, 1, 0) == 0, lib.annp_hip_last_error(h)
    assert lib.annp_hip_comm_route(h, 2, kinds, ptrs, cnts, peers, st) == 0, lib.annp_hip_last_error(h)
    torch.cuda.synchronize(dev)
    assert torch.equal(a, b)
    peers_bad = (C.c_int * 2)(0, 1)
    assert lib.annp_hip_comm_route(h, 2, kinds, ptrs, cnts, peers_bad, st) == -1          # peer outside the communicator
    assert lib.annp_hip_comm_destroy(h) == 0
    pair.close()


@pytest.mark.parametrize("world,cells", [(1, (6, 6, 6)), (1, (12, 5, 7)), (2, (8, 5, 5)), (3, (12, 5, 5)), (4, (16, 5, 6))])
def test_replanning_in_the_library_equals_the_torch_restatement(ctx, world, cells):
    """Comm::exchange + Comm::borders as library kernels (annp_hip_replan_*) against the torch restatement in
    meng_zhang_amd/domain.py, from the same state: atoms drift by up to 1.3 A per direction (out of the box, across slab faces),
    then both re-plan.  Owned atoms, ids and velocities after the migration, the wire ghosts and every periodic image (roots and
    shifts: rebuilt positions), and the plans of the two folds agree bit for bit; twice in a row, the second time from the
    first one's result."""
    torch, lib, _, dev = ctx
    from meng_zhang_amd import PairANNP
    from meng_zhang_amd.domain import SlabDomain
    x0, box = bcc(*cells, A_FE)
    xg = perturb(x0, 4, 0.05)
    vg = (2.0 * uniform_counter(xg.size, 5).reshape(xg.shape) - 1.0)
    drifts = [(2.0 * uniform_counter(xg.size, 21 + k).reshape(xg.shape) - 1.0) * 1.3 for k in range(2)]

    def program(use_hip):
        def rank_program(rank, tp):
            pair = PairANNP(1, device=0)
            pair.settings([])
            pair.coeff(["*", "*", FE_POT, "Fe"])
            pair.init_style()
            dom = SlabDomain.from_global(xg, box, (1, 1, 1), RC_LIST, dev, tp, extra={"v": vg}, hip=(lib, pair.handle) if use_hip else None)
            snaps = []
            for k in range(3):
                if k > 0:
                    dom.x[: dom.nlocal] += torch.from_numpy(drifts[k - 1][dom.ids.cpu().numpy()]).to(dev)
                    dom.replan()
                np0 = dom.nlocal + dom.nxg
                f = torch.zeros_like(dom.x)
                f[np0:] = torch.arange(3 * dom.nimg, dtype=torch.float64, device=dev).reshape(-1, 3) * 0.125 + 1.0
                keep = dom.f
                dom.f = f
                dom._rev_saved, dom._rev = dom._rev, []
                dom.reverse()                                   # the image fold alone (its plan is what is compared)
                dom._rev = dom._rev_saved
                dom.f = keep
                torch.cuda.synchronize(dev)
                root = dom.img_root32 if dom.img_root is None else dom.img_root
                snaps.append(dict(nlocal=dom.nlocal, nxg=dom.nxg, nimg=dom.nimg, migrated=dom.migrated_last, ids=dom.ids.cpu().numpy(),
                                  x=dom.x.cpu().numpy().copy(), v=dom.extra["v"].cpu().numpy().copy(), root=root.cpu().numpy().astype(np.int64),
                                  shift=dom.img_shift.cpu().numpy().copy(), fold=f[:np0].cpu().numpy().copy(),
                                  zero=float(dom.f.abs().max()) if dom.nall else 0.0))
                # the forward fill reproduces the images it was planned with
                xb = dom.x.clone()
                dom.x[np0:] = -1.0
                dom.forward()
                torch.cuda.synchronize(dev)
                assert torch.equal(dom.x[: dom.nlocal], xb[: dom.nlocal])
                assert float((dom.x[np0:] - xb[np0:]).abs().max()) < 1e-12 if dom.nimg else True
            pair.close()
            return snaps
        return ThreadFabric(world).run(rank_program)

    a, b = program(True), program(False)
    moved = 0
    for ra, rb in zip(a, b):
        for sa, sb in zip(ra, rb):
            for key in ("nlocal", "nxg", "nimg", "migrated"):
                assert sa[key] == sb[key], key
            for key in ("ids", "x", "v", "root", "shift"):
                assert np.array_equal(sa[key], sb[key]), key
            assert np.abs(sa["fold"] - sb["fold"]).max() == 0.0 or np.allclose(sa["fold"], sb["fold"], rtol=0, atol=1e-9)
            assert sa["zero"] == 0.0
            moved += sa["migrated"]
    assert moved > 0 or world == 1
