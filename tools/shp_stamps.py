#!/usr/bin/env python3
"""Developer tool (GPU box): where a wave of annp_fe_force_shp spends a step.  Needs a library built with -DANNP_SHF_STAMPS
(make -C meng_zhang_amd/csrc stamps): the kernel then writes s_memtime stamps into the descriptor rows, read back here.
   ANNP_HIP_LIBRARY=$PWD/meng_zhang_amd/libannp_hip_stamps.so python tools/shp_stamps.py 80"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

NAMES = ["requests", "geom+columns", "resolve+wait", "inserts", "table", "barrier", "flush"]


def main():
    import torch
    from annp_testlib import A_FE, FE_POT, bcc, perturb
    from meng_zhang_amd import PairANNP
    from meng_zhang_amd.domain import SlabDomain
    from meng_zhang_amd.lib import load_library
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    x0, box = bcc(n, n, n, A_FE)
    xg = perturb(x0, 12345, 0.05)
    lib = load_library()
    dev = torch.device("cuda", 0)
    dom = SlabDomain.from_global(xg, box, (1, 1, 1), 8.5, dev)
    pair = PairANNP(1, device=0)
    pair.settings([])
    pair.coeff(["*", "*", FE_POT, "Fe"])
    pair.init_style()
    h = pair.handle
    st = torch.cuda.current_stream(dev).cuda_stream
    pn, pf, pg, mx = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(0)
    assert lib.annp_hip_neigh_build_device(h, dom.nlocal, dom.nall, dom.x.data_ptr(), 8.5, C.byref(pn), C.byref(pf), C.byref(pg), C.byref(mx), st) == 0
    eng = torch.zeros(1, dtype=torch.float64, device=dev)
    for _ in range(3):
        dom.f.zero_()
        assert lib.annp_hip_compute_device(h, dom.nlocal, dom.nall, dom.x.data_ptr(), None, None, pn, pf, pg, mx.value, dom.f.data_ptr(), None, eng.data_ptr(), None, None, st) == 0
    rows = np.zeros((dom.nlocal, 32))
    assert lib.annp_hip_last_descriptors(h, rows.ctypes.data_as(C.POINTER(C.c_double)), dom.nlocal) == 0
    t = rows.view(np.uint64)                 # one row per wave and step: row (unit * 8 + group * 4 + wave-in-group)
    wq = np.arange(len(t)) % 4
    ok = t[:, 7] > t[:, 0]
    step = (t[:, 14] % np.uint64(64)).astype(np.int64)
    for name, sel in (("heavy waves (q = 0..2), steps >= 1", ok & (wq < 3) & (step >= 1)), ("light waves (q = 3), steps >= 1", ok & (wq == 3) & (step >= 1)),
                      ("heavy waves, step 0", ok & (wq < 3) & (step == 0)), ("light waves, step 0", ok & (wq == 3) & (step == 0))):
        u = t[sel]
        if len(u) == 0:
            continue
        d = (u[:, 1:8].astype(np.int64) - u[:, 0:7].astype(np.int64)).astype(np.float64)
        print("%s: %d; cycles per wave and step (mean / median / p90):" % (name, len(u)))
        for k, nm in enumerate(NAMES):
            print("  %-14s %9.1f %9.1f %9.1f" % (nm, d[:, k].mean(), np.median(d[:, k]), np.percentile(d[:, k], 90)))
        life = (u[:, 7].astype(np.int64) - u[:, 0].astype(np.int64)).astype(np.float64)
        print("  %-14s %9.1f %9.1f %9.1f" % ("step", life.mean(), np.median(life), np.percentile(life, 90)))
    life = (t[:, 7].astype(np.int64) - t[:, 0].astype(np.int64)).astype(np.float64)
    sel = ok & (wq == 0)
    print("step length by step of the run (wave 0 of a group; mean / median):")
    for k in range(int(step[sel].max()) + 1):
        m = sel & (step == k)
        if m.any():
            print("  s=%2d  %9.0f %9.0f   n=%d" % (k, life[m].mean(), np.median(life[m]), m.sum()))
    print("percentiles of the step length, steps >= 1: " + " ".join("%d:%.0f" % (q, np.percentile(life[sel & (step >= 1)], q)) for q in (1, 10, 25, 50, 75, 90, 99)))
    xcc = (t[:, 13] & np.uint64(0xf)).astype(np.int64)
    print("by XCD (mean step, steps >= 1): " + " ".join("%d:%.0f" % (k, life[sel & (step >= 1) & (xcc == k)].mean()) for k in range(8) if (sel & (xcc == k)).any()))
    # when in the kernel: start time of the step relative to the kernel's first stamp, in tenths of the span
    t0 = t[ok][:, 0].min(); span = float(t[ok][:, 7].max() - t0)
    ph = ((t[:, 0].astype(np.float64) - float(t0)) / span * 10).astype(np.int64).clip(0, 9)
    print("by tenth of the kernel's span (mean step): " + " ".join("%.0f" % life[sel & (step >= 1) & (ph == k)].mean() for k in range(10) if (sel & (step >= 1) & (ph == k)).any()))
    print("kernel span %.0f cycles" % span)
    # do waves w and w + 4 of a workgroup share a SIMD?  rows of a unit: group 0 waves 0..3, group 1 waves 0..3
    nun = len(t) // 8
    sim_all = ((t[: nun * 8, 15] >> np.uint64(4)) & np.uint64(3)).reshape(nun, 2, 4)
    okk = ok[: nun * 8].reshape(nun, 8).all(axis=1)
    same = (sim_all[okk, 0, :] == sim_all[okk, 1, :]).mean(axis=0)
    print("waves w and w+4 on the same SIMD (fraction, by w): %s;  the four waves of a group on four SIMDs: %.3f" % (
        np.round(same, 3), np.mean([len(set(r)) == 4 for r in sim_all[okk, 0, :][:20000].tolist()])))
    rot = (sim_all[okk, 0, 0].astype(np.int64)) % 4
    print("SIMD of wave 0: %s" % dict(zip(*[a.tolist() for a in np.unique(rot, return_counts=True)])))
    print("SIMD of waves 0..3 relative to wave 0's (first units): %s" % ((sim_all[okk, 0, :][:6].astype(np.int64) - sim_all[okk, 0, :1][:6].astype(np.int64)) % 4).tolist())
    tg = ((t[: nun * 8, 15] >> np.uint64(16)) & np.uint64(15)).reshape(nun, 8)[okk]
    print("TG_ID values: %s" % dict(zip(*[a.tolist() for a in np.unique(tg[:, 0], return_counts=True)])))
    hw = t[ok][:, 15]
    simd = (hw >> np.uint64(4)) & np.uint64(3)
    for q in range(4):
        v, c = np.unique(simd[(wq[ok] == q)], return_counts=True)
        print("wave-in-group %d on SIMD: %s" % (q, dict(zip(v.tolist(), c.tolist()))))


if __name__ == "__main__":
    main()
