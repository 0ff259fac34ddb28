#!/usr/bin/env python3
"""Developer tool: wall time of single MD steps of bench.py's loop (one rank), a reneighbouring step among them.
   python tools/step_times.py 40"""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import bench
    from meng_zhang_amd.domain import NoTransport
    cells = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    wl = sys.argv[2] if len(sys.argv) > 2 else "fe"
    args = types.SimpleNamespace(dt=0.001)
    dev = torch.device("cuda", 0)
    leg = bench.Leg(args, wl, cells, dev, NoTransport(), False, 0)
    leg.prime()
    for _ in range(3):
        leg.step()
    torch.cuda.synchronize(dev)
    for k in range(24):
        rebuild = k in (5, 15)
        t = time.perf_counter()
        leg.step(rebuild)
        t_issue = time.perf_counter() - t
        torch.cuda.synchronize(dev)
        print("step %2d %s issue %6.2f ms  done %6.2f ms" % (k, "REBUILD" if rebuild else "       ", t_issue * 1e3, (time.perf_counter() - t) * 1e3))
    leg.close()


if __name__ == "__main__":
    main()
