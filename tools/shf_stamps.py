#!/usr/bin/env python3
"""Developer tool (GPU box): where a wave of annp_fe_force_sh spends its life.  Needs a library built with -DANNP_SHF_STAMPS
(ANNP_HIP_LIBRARY=...): the kernel then writes s_memtime stamps into the descriptor rows, read back here.
   ANNP_HIP_LIBRARY=$PWD/meng_zhang_amd/libannp_hip_stamps.so python tools/shf_stamps.py 80"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


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
    t = rows.view(np.uint64)                 # one row per wave (the four waves of a group use the rows of its four atoms)
    ok = t[:, 8] > t[:, 0]
    t = t[ok]
    d = (t[:, 1:9].astype(np.int64) - t[:, 0:8].astype(np.int64)).astype(np.float64)
    names = ["start->count", "gather,stage", "convert", "barrier1", "turn", "tail", "barrier2", "flush"]
    print("waves %d; cycles per wave (mean / median / p90):" % len(t))
    for k, nm in enumerate(names):
        print("  %-14s %9.0f %9.0f %9.0f" % (nm, d[:, k].mean(), np.median(d[:, k]), np.percentile(d[:, k], 90)))
    life = (t[:, 8].astype(np.int64) - t[:, 0].astype(np.int64)).astype(np.float64)
    print("  %-14s %9.0f %9.0f %9.0f" % ("life", life.mean(), np.median(life), np.percentile(life, 90)))
    span = float(t[:, 8].max() - t[:, 0].min())
    print("kernel span %.0f ticks; sum of lives / span / (1024 SIMDs) = %.2f waves per SIMD" % (span, life.sum() / span / 1024))
    hw = t[:, 15]
    print("distinct (se,sh,cu,simd) slots:", len(np.unique(hw & 0xffff)))
    for nm, off, w in (("wave_id", 0, 4), ("simd", 4, 2), ("pipe", 6, 2), ("cu", 8, 4), ("sh", 12, 1), ("se", 13, 3), ("tg", 16, 4), ("vm", 20, 4), ("queue", 24, 3), ("state", 27, 3), ("me", 30, 2)):
        v, c = np.unique((hw >> np.uint64(off)) & np.uint64((1 << w) - 1), return_counts=True)
        print("  %-8s %s" % (nm, dict(zip(v.tolist(), c.tolist()))))
    # per SIMD (xcc, se, cu, simd): how much of the time are 0 / 1 / 2 waves inside their turns (stamps 4..5)?
    xcc = t[:, 13] & np.uint64(0xf)
    key = ((xcc.astype(np.int64) << 16) | (hw & np.uint64(0xfff0)).astype(np.int64))
    uk = np.unique(key)
    print("SIMDs seen: %d" % len(uk))
    tot = np.zeros(4)
    gaps = []
    for k in uk[:64]:
        m = key == k
        b0, b1 = t[m, 4].astype(np.int64), t[m, 5].astype(np.int64)
        s0, s1 = t[m, 0].astype(np.int64), t[m, 8].astype(np.int64)
        ev = sorted([(x, 1) for x in b0] + [(x, -1) for x in b1])
        cur, last = 0, ev[0][0]
        for x, d in ev:
            tot[min(cur, 3)] += x - last
            last = x
            cur += d
        st = np.sort(s0)
        gaps += list(np.diff(st))
    print("time with 0/1/2/3+ waves of a SIMD in their turns: %s" % np.round(tot / tot.sum(), 3))
    gaps = np.array(gaps)
    print("gap between consecutive wave starts on one SIMD: median %.0f, p10 %.0f, p90 %.0f (life %.0f)" % (np.median(gaps), np.percentile(gaps, 10), np.percentile(gaps, 90), np.median(life)))


if __name__ == "__main__":
    main()
