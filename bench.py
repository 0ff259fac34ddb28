#!/usr/bin/env python3
"""bench.py -- atom-steps/s of the pair_style annp hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for any N: started without a launcher and N > 1, this file starts its own N ranks (one fresh
process per GPU, before anything has touched the GPU) and relays rank 0's line.

--workload ni / anna run the same harness on BASELINE.json's config 4 (fcc Ni, 512 000 atoms) and on
pair_style anna_adp; the default is the metric's own workload:

Workload (BASELINE.json): bcc-Fe ANNP, 80x80x80 cells x 2 = 1 024 000 atoms, fully
periodic, a = 2.8553 A, every coordinate displaced by U(-0.05, 0.05) A from a
counter-based generator, neighbour list cutoff 6.5 + 2.0 A.  With N ranks the box is
cut into N slabs along x (strong scaling: total work fixed); ghosts travel by
point-to-point halo exchange over RCCL (meng_zhang_amd/domain.py).

One step = what one MD step asks of the path: positions -> ghosts (forward halo),
one force evaluation of every owned atom (descriptor pass, FP64-MFMA network pass,
force pass), ghost forces -> owners (reverse halo), and the velocity-Verlet update of
the owned atoms; the total energy is all-reduced every `--thermo` steps.  Inputs are
resident in HBM before the timed region; the neighbour list is built once on the device
before it (the `mini_md` figure re-homes atoms and rebuilds ghosts + list every 10 steps).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the force pass),
timed with HIP events on its own stream inside the timed region; `cpu_baseline` is the
CPU oracle ("port" of the reference CPU pair style) timed on this host on a bounded
sample of the same workload.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# Per-unit algorithmic work (fp64, FMA = 2 flop, mul / add = 1), counted from the kernels as they execute (DESIGN.md 4.5:
# "kernel-derived", not SURVEY.md 8d's budget for the reference formulation, which is printed beside it).
# (a) what the implemented formulation needs -- the roofline numerator.  The Chebyshev passes work on the 361 moments of a
#     neighbourhood (fe_sh_kernels.hpp, fe_shf_kernels.hpp), not on its pairs.
#     Descriptor pass (round 4b: monomial moments), per in-cutoff neighbour: the powers of z of the 19 columns (K - 2 multiplies in
#     a column of K powers: 1 and z cost nothing; round 6: columns (10,11), (12,13), (14,15), (16,17) share theirs -- 6 + 4 + 2 + 0
#     multiplies fewer; columns (5,6) and (7,8) do too, each in two sweeps over the neighbours: 15 multiplies fewer, the second sweep's
#     fc (x+iy)^(m+1) made again for 6 flop each) 153 - 12 - 15 + 12 = 138; the accumulation behind every power (1 FMA in column 0, 2 elsewhere)
#     2 x 19 + 4 x 171 = 722; advancing the power (x+iy)^m 18 x 6 = 108; geometry, cutoff function, 9 radial functions 110.
#     Per atom: lane sums 361 x 15, 19 x 19 product ~ 7 k; the change of basis to the moments of the Pm^(m)_k 1 430 FMAs, power
#     spectrum 3 x 190.
#     Force pass (round 4: Horner's rule on monomial coefficients), per in-cutoff neighbour: columns of K >= 3 entries start with
#     6 FMAs for three entries and take 4 FMAs per entry after that (16 x 6 + 4 x 120 = 576 FMAs), the two-entry column 2, the
#     cosine-only column m = 0 3 + 2 x 16 = 35, Horner's rule in w 12 FMAs per column end (17 of them: behind m = 17 .. 1) and 8 at the end
#     (212): 825 FMAs = 1 650 flop;
#     geometry, radial T and T', force assembly 158.  Per atom: B = W kappa A 760, change of basis 1 430 FMAs = 2 860.
FLOP_PAIR_DESC, FLOP_NBR_DESC, FLOP_ATOM_DESC = 0.0, 138 + (2 * 19 + 4 * 171) + 18 * 6 + 110.0, 7000.0 + 2 * 1430 + 3 * 190
FLOP_PAIR_FORCE, FLOP_NBR_FORCE, FLOP_ATOM_FORCE = 0.0, 2 * (16 * 6 + 4 * 120 + 2 + 35 + 12 * 17 + 8) + 158.0, 760.0 + 2860.0
#     ... and the pair-loop kernels they replaced (ANNP_HIP_FE_DESC=pairs ANNP_HIP_FE_FORCE=pairs, or a system with more than
#     128 neighbours per atom): pass 1: cos 5, weights 3, T_2..T_18 recurrence + accumulate 68; pass 3: cos 5, Horner P 36 +
#     dP 34, dP fc_b 1, a-side 8, b-side 4
PAIRLOOP_FLOP = {"desc": (76.0, 110.0, 0.0), "force": (88.0, 158.0, 0.0)}
FLOP_MLP = 2500.0
# (b) SURVEY.md 8d's budget for the reference formulation (T and T' recurrences for every function in both
#     passes): 350 flop per pair + 175 per neighbour + 1.6 k = 2.20 MFLOP per atom-step at n = 112
SURVEY_FLOP_PAIR, SURVEY_FLOP_NBR, SURVEY_FLOP_MLP = 350.0, 175.0, 1600.0
BYTES_ATOM_STEP = 9960.0 + 2 * 380 * 8            # gathered bytes per atom-step + the moments written and read back
PEAK_FP64_VECTOR = 78.6                           # TFLOP/s, MI355X (MI355X_MICROARCH.md: half of FP32 vector 157.3)
PEAK_HBM = 8000.0                                 # GB/s spec
# Ni (Behler G2/G4), counted from the kernels: per candidate (j,k) pair the distance pre-pass of the descriptor
# pass (squared distance and three compares: 12; the force pass reads the surviving pairs from the list that pass leaves);
# per pair that survives r_ij, r_ik, r_jk < Rc (39 % of them in fcc Ni, the figure used here) geometry + cutoff function of
# r_jk 75, one exp and its powers 44, squaring ladders 10, the 8 (lambda, zeta) steps 64 (descriptor) / 136 (force), force
# assembly 57: 195 + 320; per in-range neighbour sincos + one exp and its powers in each pass 300; network 27-24-24-1 forward +
# reverse 5 k.  (SURVEY.md 8d's 24 x 40 + 150 per candidate pair priced a pow() per function and every candidate pair.)
NI_IN_RANGE_PAIR_FRACTION = 0.39
NI_FLOP_PAIR = 12.0 + NI_IN_RANGE_PAIR_FRACTION * (195.0 + 320.0)
NI_FLOP_NBR, NI_FLOP_MLP = 300.0, 5000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cells", type=int, default=80, help="bcc cells per edge (80 -> 1 024 000 atoms)")
    ap.add_argument("--cpu-sample", type=int, default=65536, help="atoms in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--dt", type=float, default=0.001, help="ps")
    ap.add_argument("--workload", choices=["fe", "ni", "anna"], default="fe",
                    help="fe = the BASELINE.json metric (default); ni = config 4 (fcc Ni, 40x40x80 cells = 512 000 atoms, "
                         "cells taken as cells/2 x cells/2 x cells); anna = pair_style anna_adp on the bcc-Fe box")
    ap.add_argument("--rebuild-every", type=int, default=10,
                    help="secondary figure: the same steps with atoms re-homed, ghosts re-derived and the neighbour list rebuilt "
                         "on the device every N steps (0 = skip)")
    ap.add_argument("--thermo", type=int, default=10, help="all-reduce the total energy every N steps (LAMMPS `thermo N`)")
    ap.add_argument("--secondary", type=int, default=1,
                    help="1: after the metric's own workload also run BASELINE.json's fcc-Ni configuration (512 000 atoms, 10 steps) "
                         "and report it as secondary.ni (single GPU, default workload only); 0 = skip")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, one fresh process
    per GPU under torch.distributed.run, BEFORE anything in this process has touched the GPU (this parent never does,
    it does not even import torch), relay rank 0's JSON line and return the children's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        out = out.strip()
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 or line is not None else 1


class _HostStaged:
    """Rehearsal transport (ANNP_BENCH_BACKEND=gloo with the ranks sharing one card): TorchTransport's surface with device
    buffers bounced through host memory, because gloo has no device point-to-point.  Never the measured configuration:
    it exists so that the N > 1 control flow of this file can run with the real force engine where only one GPU is visible."""

    def __init__(self, dist):
        from meng_zhang_amd.domain import TorchTransport
        self.inner = TorchTransport(dist)
        self.world, self.rank = self.inner.world, self.inner.rank

    def route(self, msgs):
        staged, back = [], []
        for kind, t, peer in msgs:
            h = t.detach().cpu() if kind == "send" else t.new_empty(t.shape, device="cpu")
            staged.append((kind, h, peer))
            if kind == "recv":
                back.append((t, h))
        self.inner.route(staged)
        for t, h in back:
            t.copy_(h)

    def allgather(self, t):
        return self.inner.allgather(t.cpu())

    def allreduce_max(self, v):
        return self.inner.allreduce_max(v.cpu())

    def allreduce_sum_(self, t):
        h = t.cpu()
        self.inner.allreduce_sum_(h)
        t.copy_(h)
        return t


class Leg:
    """One workload resident on this rank's GPU: the slab of the box, the pair style, the device-built list, and
    step() = one MD step in LAMMPS' Verlet::run order.  The step's own jobs (integrator halves, halo pack, image fill +
    force clear, ghost-force fold) are the library's kernels (annp_hip_verlet_half / _halo_pack / _halo_unpack_images /
    _reverse_fold); torch only owns the buffers and the RCCL point-to-point group."""

    def __init__(self, args, wl, cells, dev, tp, dry, local_rank, wire_self=False, wire_lib=False):
        import torch
        from meng_zhang_amd.workloads import A_FE, A_NI, ANNA_POT, FE_POT, NI_POT, bcc, fcc, load_fe_st, perturb
        from meng_zhang_amd.domain import SlabDomain
        self.args, self.wl, self.dev, self.dry, self.torch = args, wl, dev, dry, torch
        self.rc_list = 7.055 if wl == "anna" else 8.5
        periodic = (1, 1, 1)
        if wl == "fe_st":                   # the reference's published deck: its own data file, `boundary m p m`
            xg, box = load_fe_st()
            periodic = (0, 1, 0)
            if os.environ.get("ANNP_BENCH_FE_ST_ORDER", "lammps") == "lammps":
                # LAMMPS sorts atoms in space when a run is set up (Verlet::setup -> Atom::sort, atom_modify sort's default) and every
                # 1000 steps after: the order its pair styles see is that one, not the data file's
                from meng_zhang_amd.workloads import lammps_sort_order
                xg = xg[lammps_sort_order(xg, box)]
        else:
            if wl == "ni":
                x0, box = fcc(cells // 2, cells // 2, cells, A_NI)
            else:
                x0, box = bcc(cells, cells, cells, A_FE)
            xg = perturb(x0, 12345, 0.05)
        self.natoms = xg.shape[0]
        self.potfile, element, style, self.mass = {"fe": (FE_POT, "Fe", "annp", 55.847), "ni": (NI_POT, "Ni", "annp", 58.6934),
                                                   "anna": (ANNA_POT, "Fe", "anna_adp", 55.847), "fe_st": (FE_POT, "Fe", "annp", 55.847)}[wl]
        self.lib = self.h = self.pair = None
        if not dry:
            from meng_zhang_amd import PairANNP
            from meng_zhang_amd.lib import load_library
            self.lib = load_library()
            self.pair = PairANNP(ntypes=1, device=local_rank, style=style)
            self.pair.settings([])
            self.pair.coeff(["*", "*", self.potfile, element])
            self.pair.init_style()
            self.h = self.pair.handle
        hip = None if (dry or os.environ.get("ANNP_BENCH_TORCH_STEP") == "1") else (self.lib, self.h)
        if wire_lib:
            from meng_zhang_amd.domain import LibTransport
            tp = LibTransport(self.lib, self.h, tp, dev)
        self.dom = SlabDomain.from_global(xg, box, periodic, self.rc_list, dev, tp, extra={"v": np.zeros_like(xg)}, hip=hip, wire_self=wire_self)
        self.p_num, self.p_first, self.p_neigh, self.mx = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int(0)
        self.eng = torch.zeros(1, dtype=torch.float64, device=dev)
        self.vir = torch.zeros(6, dtype=torch.float64, device=dev) if wl == "fe_st" else None       # vflag_global of every NPT step
        ftm2v = 1.0 / 1.0364269e-4          # LAMMPS metal units: (eV/A)/(g/mol) -> A/ps^2
        self.dtf = 0.5 * args.dt * ftm2v / self.mass
        self.reissued = 0
        self.halo_marks = None              # when a list: (before, between, after) marks of the two exchanges of every step
        self.mark_pool = []
        self.build_list()

    def stream(self):
        return self.torch.cuda.current_stream(self.dev).cuda_stream

    def check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.annp_hip_last_error(self.h).decode()))

    def build_list(self):
        """neighbour list on the device (annp_gpu_compute_n analogue), from resident positions"""
        if not self.dry:
            d = self.dom
            self.check(self.lib.annp_hip_neigh_build_device(self.h, d.nlocal, d.nall, d.x.data_ptr(), self.rc_list, C.byref(self.p_num),
                                                            C.byref(self.p_first), C.byref(self.p_neigh), C.byref(self.mx), self.stream()),
                       "neigh_build")

    def force_eval(self):
        """Pair::compute on cleared f / eng"""
        if self.dry:
            return
        d = self.dom
        if self.vir is not None:
            self.vir.zero_()
        for attempt in range(2):
            rc = self.lib.annp_hip_compute_device(self.h, d.nlocal, d.nall, d.x.data_ptr(), None, None, self.p_num, self.p_first,
                                                  self.p_neigh, self.mx.value, d.f.data_ptr(), None, self.eng.data_ptr(),
                                                  self.vir.data_ptr() if self.vir is not None else None, None, self.stream())
            if rc == -7 and attempt == 0:       # a deferred Behler capacity error: the capacity has been raised, issue this one again
                self.reissued += 1
                continue
            self.check(rc, "compute_device")
            return

    def spin_up(self, min_ms):
        """Force evaluations (no MD step: positions do not change, f is cleared by prime() afterwards) until the device has been busy for
        min_ms: after the host-bound set-up the chip takes 20-30 ms of work to reach the clocks it then holds (measured at 128 000
        atoms: the first 10 steps behind 2 warm-up steps ran 1.52 ms each, the same steps behind 20 warm-up steps 1.37), and W warm-up
        steps of a small system are over before that.  Returns the number of evaluations issued (reported in config.spin_up)."""
        if self.dry or min_ms <= 0:
            return 0
        n, t0 = 0, time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < min_ms and n < 400:
            for _ in range(4):
                self.force_eval()
            n += 4
            self.torch.cuda.synchronize(self.dev)
        return n

    def prime(self):
        self.dom.forward(clear_forces=True, eng=self.eng)
        self.force_eval()
        self.dom.reverse()

    def step(self, rebuild=False):
        """LAMMPS Verlet::run order: initial_integrate; on a reneighbouring step exchange + borders + neighbour build,
        otherwise forward_comm; force_clear, pair compute, reverse_comm; final_integrate"""
        a, d = self.args, self.dom
        d.verlet_half(d.extra["v"], self.dtf, a.dt)         # FixNVE::initial_integrate
        m0 = self.mark()
        if rebuild:
            d.replan(eng=self.eng)          # Comm::exchange + Comm::borders (atoms, velocities and ids change rank here); f and the energy word come back zeroed
            self.build_list()               # Neighbor::build
        else:
            d.forward(clear_forces=True, eng=self.eng)      # Comm::forward_comm + Verlet::force_clear
        m1 = self.mark()
        self.force_eval()                   # Pair::compute
        m2 = self.mark()
        d.reverse()                         # Comm::reverse_comm
        m3 = self.mark()
        if self.halo_marks is not None:
            self.halo_marks.append((m0, m1, m2, m3))
        d.verlet_half(d.extra["v"], self.dtf, 0.0)          # FixNVE::final_integrate

    def mark(self):
        """a point in time on the step's stream (an event) -- or on the host clock in the CPU rehearsal"""
        if self.halo_marks is None:
            return None
        if self.dry:
            return time.perf_counter()
        if not self.mark_pool:              # (the events are made before the timed region, ADVICE r4: making four per step inside it is
            ev = self.torch.cuda.Event(enable_timing=True)      # host time that the N = 8 steps of 1.5 ms would feel)
        else:
            ev = self.mark_pool.pop()
        ev.record(self.torch.cuda.current_stream(self.dev))
        return ev

    def reserve_marks(self, steps):
        """events for `steps` steps of marks, made ahead of the timed region"""
        if not self.dry:
            self.mark_pool = [self.torch.cuda.Event(enable_timing=True) for _ in range(4 * steps)]

    def halo_ms(self):
        """mean milliseconds per step in the forward and in the reverse exchange (device time between the marks)"""
        if not self.halo_marks:
            return 0.0, 0.0
        fw = bw = 0.0
        for m0, m1, m2, m3 in self.halo_marks:
            if self.dry:
                fw += (m1 - m0) * 1e3; bw += (m3 - m2) * 1e3
            else:
                fw += m0.elapsed_time(m1); bw += m2.elapsed_time(m3)
        n = len(self.halo_marks)
        return fw / n, bw / n

    def timed(self, steps, warmup):
        """(seconds, HIP-event means of the kernels) of `steps` steps; single rank, no thermo"""
        torch = self.torch
        self.spin_up(float(os.environ.get("ANNP_BENCH_SPINUP_MS", "60")))
        self.prime()
        for _ in range(warmup):
            self.step()
        torch.cuda.synchronize(self.dev)
        self.check(self.lib.annp_hip_set_timing(self.h, 1), "set_timing")
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize(self.dev)
        dt = time.perf_counter() - t0
        ms4, ns = np.zeros(4), C.c_int(0)
        self.check(self.lib.annp_hip_timing_stats(self.h, ms4.ctypes.data_as(C.POINTER(C.c_double)), C.byref(ns)), "timing_stats")
        self.lib.annp_hip_set_timing(self.h, 0)
        self.check(self.lib.annp_hip_sync(self.h), "sync (deferred device-side errors of the timed steps)")
        return dt, ms4, int(ns.value)

    def counts(self):
        c = np.zeros(self.dom.nlocal, dtype=np.int32)
        self.check(self.lib.annp_hip_last_counts(self.h, c.ctypes.data_as(C.POINTER(C.c_int)), self.dom.nlocal), "last_counts")
        return c.astype(np.float64)

    def close(self):
        if self.pair is not None:
            self.pair.close()
            self.pair = None


def ni_flops(n):
    """algorithmic flop of one Behler evaluation from the per-atom in-range counts n"""
    pairs, nbrs = float((n * (n - 1) / 2).sum()), float(n.sum())
    return pairs * NI_FLOP_PAIR + nbrs * NI_FLOP_NBR + n.size * NI_FLOP_MLP


def secondary_ni(args, dev, local_rank):
    """BASELINE.json config 5 (fcc Ni, 40x40x80 cells = 512 000 atoms, the Behler G2/G4 potential) as a short extra leg of
    the default run, so that it gets a driver-timed line too: same step, 10 steps after 2 warm-up steps."""
    from meng_zhang_amd.domain import NoTransport
    leg = Leg(args, "ni", 80, dev, NoTransport(), False, local_rank)
    steps, warmup = 10, 2
    dt, ms4, ns = leg.timed(steps, warmup)
    n = leg.counts()
    flop = ni_flops(n)
    ev = flop / (float(ms4[3]) * 1e-3) / 1e12
    e = float(leg.eng.item())
    lc = leg.lib.annp_hip_list_cutoff(leg.h, leg.rc_list)
    out = {"workload": "%d-atom fcc-Ni ANNP (40x40x80 cells x4, a=3.52, +-0.05 A displacements), ni_annp_potential_2.ann, "
                       "list asked for at 8.5 A, built at %.3f A (descriptor cutoff 3.9 A + the 2 A skin)" % (leg.natoms, lc),
           "value": leg.natoms * steps / dt, "unit": "atom-steps/s", "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
           "kernel_ms": {"descriptor": float(ms4[0]), "network": float(ms4[1]), "force": float(ms4[2]), "evaluation": float(ms4[3]), "samples": ns},
           "roofline": {"kernel": "annp_ni_desc + annp_mlp_mfma + annp_ni_force (whole evaluation)", "bound": "fp64_valu", "achieved": ev,
                        "peak": PEAK_FP64_VECTOR, "unit": "TFLOP/s", "frac": ev / PEAK_FP64_VECTOR, "traffic": None,
                        "algorithmic_flop_per_launch": flop,
                        "flop_per_unit": {"candidate_pair": NI_FLOP_PAIR, "neighbour": NI_FLOP_NBR, "atom": NI_FLOP_MLP}},
           "neighbors_in_cutoff_mean": float(n.mean()), "list_neighbors_max": int(leg.mx.value), "energy_per_atom": e / leg.natoms,
           "evaluations_reissued": leg.reissued}
    if args.cpu_sample > 0:
        try:
            from annp_testlib import FAST, KIND_NI_FIXED, NI_POT, oracle_compute, oracle_lib, read_pot
            m = min(args.cpu_sample, leg.dom.nlocal)
            s = _sample_system(leg.lib, leg.h, leg.dom.x.cpu().numpy(), leg.dom.nlocal, leg.dom.nall, m, leg.rc_list)
            nthreads = min(oracle_lib().annp_oracle_max_threads(), _cpu_share())
            pot = read_pot(NI_POT)
            oracle_compute(pot, s, KIND_NI_FIXED, FAST, inum=min(m, 256), nthreads=nthreads)
            t1 = time.perf_counter()
            oracle_compute(pot, s, KIND_NI_FIXED, FAST, inum=m, nthreads=nthreads)
            tc = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": m / tc, "unit": "atom-steps/s", "cores": int(nthreads), "kind": "port", "seconds": tc,
                                   "sample": "1 force evaluation of the first %d of %d atoms of the same box (list at 8.5 A, as the reference "
                                             "would be given), oracle FAST strategy with the GPU kernel's derivative (Ni-fixed), OpenMP" % (m, leg.natoms)}
        except Exception as exc:
            out["cpu_baseline"] = {"error": repr(exc)}
    leg.close()
    return out


FE_ST_LOG_VOLUME = 1773495.9        # A^3: what LAMMPS printed for fe_st.dat under `boundary m p m` (the reference's log, thermo "Volume", step 0)


def secondary_fe_st(args, dev, local_rank):
    """The reference's own published deck as a short extra leg: fe_st.dat, 152 880 Fe atoms, `boundary m p m`
    (free surfaces in x and z, ragged neighbour counts, atoms in the data file's id order), the global virial tallied every step
    as the deck's `fix npt` has it, list built on the device.  BASELINE.md's only published throughput belongs to this
    file: 85.4 k atom-steps/s on 2 GPUs of the reference's mixed-precision build (zip log_relaxing_new.lammps:1168-1176,
    NPT, 1000 steps) -- another machine, another precision, an NPT integrator this leg does not run (it steps NVE with the
    virial on): the figure stands beside the leg's, it is not a ratio."""
    from meng_zhang_amd.domain import NoTransport
    leg = Leg(args, "fe_st", 0, dev, NoTransport(), False, local_rank)
    steps, warmup = 10, 2
    dt, ms4, ns = leg.timed(steps, warmup)
    n = leg.counts()
    e, vir = float(leg.eng.item()), leg.vir.cpu().numpy()
    out = {"workload": "fe_st.dat of the reference's performance test (%d Fe atoms, boundary m p m, atoms in %s), fe_annp_potential_2.ann, "
                       "device-built list at 8.5 A, global virial every step, NVE dt = %g ps" % (
                           leg.natoms, "the order LAMMPS' set-up sort leaves (atom_modify sort: bins of 4.25 A)" if os.environ.get("ANNP_BENCH_FE_ST_ORDER", "lammps") == "lammps" else "the data file's id order", args.dt),
           "value": leg.natoms * steps / dt, "unit": "atom-steps/s", "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
           "kernel_ms": {"descriptor": float(ms4[0]), "network": float(ms4[1]), "force": float(ms4[2]), "evaluation": float(ms4[3]), "samples": ns},
           "neighbors_in_cutoff_mean": float(n.mean()), "neighbors_in_cutoff_max": int(n.max()), "list_neighbors_max": int(leg.mx.value),
           "energy_last_step_eV": e, "pressure_virial_last_step_bar": float(vir[:3].sum() / (3 * FE_ST_LOG_VOLUME) * 1.6021765e6),
           "pressure_volume_A3": FE_ST_LOG_VOLUME, "pressure_volume_source": "thermo column Volume at step 0 of the reference's log_relaxing_new.lammps for this deck "
           "(boundary m p m: LAMMPS' shrink-wrapped box, not derivable from the data file's 1773141.3 A^3 alone)",
           "eval_path": int(leg.lib.annp_hip_eval_path(leg.h)),
           "reference_published": {"value": 85.4e3, "unit": "atom-steps/s", "n_gpus": 2, "precision": "mixed (reference GPU build)",
                                   "run": "fix npt, 1000 steps", "source": "performance test.zip log_relaxing_new.lammps:1168-1176 (BASELINE.md)",
                                   "note": "other hardware, other precision, NPT: shown beside this leg, no ratio is formed"}}
    leg.close()
    return out


def main():
    run_rank(parse_args())


def run_rank(args):
    import torch
    import torch.distributed as dist
    from meng_zhang_amd.workloads import A_FE, A_NI, ANNA_POT, FE_POT, NI_POT, bcc, fcc, perturb
    from meng_zhang_amd.domain import NoTransport, SlabDomain, TorchTransport

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: start it as `python bench.py --gpus N` or under a launcher with "
                         "--nproc-per-node N\n" % (args.gpus, world))
        raise SystemExit(2)
    # ANNP_BENCH_DRYRUN=1: rehearsal of the control flow where there is no GPU at all (the launcher test): ranks on CPU over
    # gloo, every step of the loop except the force evaluation itself, no throughput reported.  There is no CPU force path.
    dry = os.environ.get("ANNP_BENCH_DRYRUN") == "1"
    staged = os.environ.get("ANNP_BENCH_BACKEND") == "gloo" and not dry
    if dry:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
        if os.environ.get("ANNP_BENCH_SHARE_GPU") == "1":      # rehearsal only: several ranks on one card
            local_rank = local_rank % torch.cuda.device_count()
        elif torch.cuda.device_count() <= local_rank:
            sys.stderr.write("bench.py: rank %d (local rank %d) sees %d HIP device(s): one GPU per rank is required "
                             "(ANNP_BENCH_SHARE_GPU=1 is the rehearsal switch for ranks sharing a card)\n" % (rank, local_rank, torch.cuda.device_count()))
            raise SystemExit(3)
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("ANNP_FORCE_DIST") == "1"      # the latter: rehearse RCCL with one rank
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        import datetime
        # a wire that hangs must end the run by itself, with one line on stderr, not at the driver's limit:
        # the collectives get a timeout, and until the first whole step has run a watchdog stands behind everything
        # (set-up exchanges of the decomposition, the first point-to-point group, the first all-reduce)
        wire_timeout = float(os.environ.get("ANNP_BENCH_WIRE_TIMEOUT", "240"))
        tmo = datetime.timedelta(seconds=wire_timeout)
        if dry or staged:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        import threading

        def _wire_hung():
            sys.stderr.write("bench.py: rank %d of %d: the first exchanges over the %s backend did not complete within %.0f s "
                             "(ANNP_BENCH_WIRE_TIMEOUT): giving up\n" % (rank, world, "gloo" if (dry or staged) else "nccl (RCCL)", wire_timeout))
            sys.stderr.flush()
            os._exit(4)
        watchdog = threading.Timer(wire_timeout, _wire_hung)
        watchdog.daemon = True
        watchdog.start()
        if os.environ.get("ANNP_BENCH_TEST_HANG_RANK") == str(rank):       # (tests/test_bench_launcher.py: a rank that never answers)
            time.sleep(1e6)
        if dist.get_world_size() != args.gpus and not (world == 1 and args.gpus == 1):
            sys.stderr.write("bench.py: the process group has %d ranks, --gpus says %d\n" % (dist.get_world_size(), args.gpus))
            raise SystemExit(2)
    # ANNP_BENCH_WIRE_SELF=1 (with ANNP_FORCE_DIST=1, one rank): the x-periodic images travel through the transport to this same
    # rank instead of being local copies -- ncclSend / ncclRecv on device buffers, executed on a single GPU
    wire_self = world == 1 and use_dist and os.environ.get("ANNP_BENCH_WIRE_SELF") == "1"
    tp = (TorchTransport(dist) if wire_self else NoTransport()) if world == 1 else (_HostStaged(dist) if staged else TorchTransport(dist))

    def sync():
        if not dry:
            torch.cuda.synchronize(dev)

    # ANNP_BENCH_WIRE=lib: the point-to-point groups of the halo are issued by the library itself (annp_hip_comm_route: RCCL from
    # C++ on the compute stream) instead of torch.distributed's batch_isend_irecv.  Opt-in: the default wire is torch's.
    wire_lib = os.environ.get("ANNP_BENCH_WIRE") == "lib" and use_dist and not dry and not staged and (world > 1 or wire_self)
    leg = Leg(args, args.workload, args.cells, dev, tp, dry, local_rank, wire_self, wire_lib)
    dom, lib, h = leg.dom, leg.lib, leg.h
    wl, natoms, rc_list, potfile, mx = leg.wl, leg.natoms, leg.rc_list, leg.potfile, leg.mx
    check, step, force_eval = leg.check, leg.step, leg.force_eval
    eng = leg.eng
    e_thermo = torch.zeros(1, dtype=torch.float64, device=dev)
    nstep = [0]

    def thermo():                           # total E_pair of the current step, as LAMMPS prints it every `thermo` steps
        e_thermo.copy_(eng)
        if use_dist:
            tp.allreduce_sum_(e_thermo) if world > 1 else dist.all_reduce(e_thermo)

    def step_thermo(rebuild=False):
        step(rebuild)
        nstep[0] += 1
        if args.thermo > 0 and nstep[0] % args.thermo == 0:
            thermo()

    def barrier():
        sync()
        if use_dist:
            dist.barrier()
        sync()

    spun = leg.spin_up(float(os.environ.get("ANNP_BENCH_SPINUP_MS", "60")))
    leg.prime()
    for _ in range(args.warmup):
        step_thermo()
    thermo()                            # (an all-reduce before the timed region, whatever --thermo is)
    barrier()
    if use_dist:
        watchdog.cancel()               # every kind of exchange of a step has completed once
    if not dry:
        check(lib.annp_hip_set_timing(h, 1), "set_timing")
    leg.reserve_marks(args.steps)
    leg.halo_marks = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_thermo()
    sync()
    dt_own = time.perf_counter() - t0       # this rank's own clock, before the barrier (per_rank.step_ms)
    barrier()
    dt_wall = time.perf_counter() - t0
    halo_fw_ms, halo_bw_ms = leg.halo_ms()
    leg.halo_marks = None
    ms4 = np.zeros(4)
    ns = C.c_int(0)
    if not dry:
        check(lib.annp_hip_timing_stats(h, ms4.ctypes.data_as(C.POINTER(C.c_double)), C.byref(ns)), "timing_stats")
        lib.annp_hip_set_timing(h, 0)
        check(lib.annp_hip_sync(h), "sync (deferred device-side errors of the timed steps)")
    if use_dist:
        t = torch.tensor([dt_wall], dtype=torch.float64, device=dev)
        dt_wall = tp.allreduce_max(t) if world > 1 else float(t.item())
    thermo()
    e_total = float(e_thermo.item())
    nlocal, nall = dom.nlocal, dom.nall
    halo_rank = dom.bytes_per_exchange + dom.nxg * 24        # bytes this rank sends per step: positions out, ghost forces back

    # ---- algorithmic work of this rank's launches (actual in-cutoff counts) ------------
    counts = np.zeros(nlocal, dtype=np.int32)
    if not dry:
        check(lib.annp_hip_last_counts(h, counts.ctypes.data_as(C.POINTER(C.c_int)), nlocal), "last_counts")
    info = (C.c_int * 4)()
    if not dry:
        check(lib.annp_hip_eval_info(h, info), "eval_info")

    # ---- secondary figure (SURVEY.md 8d ii): the same MD steps with periodic re-homing + device-side list rebuilds ----
    md_rate, md_migrated = None, 0
    if args.rebuild_every > 0:
        barrier()
        t1 = time.perf_counter()
        for k in range(args.steps):
            rebuild = k % args.rebuild_every == 0
            step_thermo(rebuild)
            dom = leg.dom
            if rebuild:
                md_migrated += dom.migrated_last
        barrier()
        t_md = time.perf_counter() - t1
        if use_dist:
            t = torch.tensor([t_md], dtype=torch.float64, device=dev)
            t_md = tp.allreduce_max(t) if world > 1 else float(t.item())
        md_rate = natoms * args.steps / t_md
        if not dry:
            check(lib.annp_hip_sync(h), "sync (mini-MD steps)")

    # per-rank facts, gathered on every rank
    mine = torch.tensor([nlocal, nall - nlocal, halo_rank, md_migrated], dtype=torch.int64, device=dev)
    per_rank = tp.allgather(mine).numpy().reshape(world, 4)

    # ... and what tells a slow rank from a slow wire when the curve over N is not what it should be
    mine_f = torch.tensor([float(ms4[0]), float(ms4[1]), float(ms4[2]), float(ms4[3]), halo_fw_ms, halo_bw_ms, dt_own / args.steps * 1e3],
                          dtype=torch.float64, device=dev)
    per_rank_f = tp.allgather(mine_f).numpy().reshape(world, 7)

    n = counts.astype(np.float64)
    pairs = float((n * (n - 1) / 2).sum())
    nbrs = float(n.sum())
    pair_loop = (os.environ.get("ANNP_HIP_FE_FORCE") == "pairs" or os.environ.get("ANNP_HIP_FE_DESC") == "pairs" or
                 (n.size and n.max() > 128))
    fpd, fnd, fad = PAIRLOOP_FLOP["desc"] if (pair_loop or os.environ.get("ANNP_HIP_FE_DESC") == "pairs") else (FLOP_PAIR_DESC, FLOP_NBR_DESC, FLOP_ATOM_DESC)
    fpf, fnf, faf = PAIRLOOP_FLOP["force"] if pair_loop else (FLOP_PAIR_FORCE, FLOP_NBR_FORCE, FLOP_ATOM_FORCE)
    flop_force = pairs * fpf + nbrs * fnf + nlocal * faf
    flop_desc = pairs * fpd + nbrs * fnd + nlocal * fad
    flop_eval = flop_force + flop_desc + nlocal * FLOP_MLP
    flop_survey = pairs * SURVEY_FLOP_PAIR + nbrs * SURVEY_FLOP_NBR + nlocal * SURVEY_FLOP_MLP
    if wl == "ni":
        flop_eval = pairs * NI_FLOP_PAIR + nbrs * NI_FLOP_NBR + nlocal * NI_FLOP_MLP
        flop_survey = pairs * (24 * 40.0 + 150.0) + nlocal * NI_FLOP_MLP
    force_ms, desc_ms, mlp_ms = float(ms4[2]), float(ms4[0]), float(ms4[1])

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    value = natoms * args.steps / dt_wall
    label = {"fe": "bcc-Fe ANNP", "ni": "fcc-Ni ANNP (BASELINE.json config 4)", "anna": "bcc-Fe ANNA-ADP (pair_style anna_adp, SURVEY 8f.4)"}[wl]
    out = {
        "metric": "atom-steps/sec (whole node), %s, 1/2/4/8 MI355X" % label,
        "value": None if dry else value,
        "unit": "atom-steps/s",
        "n_gpus": world,
        "measured_n": [world],
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt_wall / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": ("%d-atom bcc-Fe ANNP (%d^3 cells x2, a=2.8553, +-0.05 A displacements), "
                         "fe_annp_potential_2.ann, list cutoff 8.5 A, x-slab decomposition" % (natoms, args.cells)) if wl == "fe" else
                        "%d-atom %s, %s, +-0.05 A displacements, list cutoff %.3f A, x-slab decomposition" % (
                            natoms, label, os.path.basename(potfile), rc_list),
            "atoms": natoms,
            "world_size": dist.get_world_size() if use_dist else 1,
            "backend": ("gloo (rehearsal)" if (dry or staged) else "nccl (RCCL)") if use_dist else "none (single rank)",
            "wire_self": bool(wire_self), "wire": "libannp_hip (annp_hip_comm_route: RCCL from C++)" if wire_lib else ("torch.distributed" if use_dist else "none"),
            "atoms_rank": [int(v) for v in per_rank[:, 0]], "ghosts_rank": [int(v) for v in per_rank[:, 1]],
            "halo_bytes_per_step": int(per_rank[:, 2].sum()),
            "neighbors_in_cutoff_mean": float(n.mean()) if not dry else None, "list_neighbors_max": int(mx.value),
            "force_pass_capacity": int(info[2]), "atoms_through_fixup_launch": int(info[1]),
            "parallelism": "spatial x%d, halo p2p" % world,
            "step": "verlet + forward halo + force evaluation + reverse halo; total energy all-reduced every %d steps" % args.thermo,
        },
        "per_rank": {"kernel_ms": {"descriptor": [float(v) for v in per_rank_f[:, 0]], "network": [float(v) for v in per_rank_f[:, 1]],
                                   "force": [float(v) for v in per_rank_f[:, 2]], "evaluation": [float(v) for v in per_rank_f[:, 3]]},
                     "halo_ms": {"forward": [float(v) for v in per_rank_f[:, 4]], "reverse": [float(v) for v in per_rank_f[:, 5]],
                                 "note": "device time per step between events around the exchange (pack / image fill + force clear and the "
                                         "wire; fold of the returned ghost forces): includes waiting for the slower neighbour"},
                     "step_ms": [float(v) for v in per_rank_f[:, 6]]},
        "energy_per_atom_eV": e_total / natoms,
        "mini_md": None if md_rate is None else {
            "value": None if dry else md_rate, "unit": "atom-steps/s", "atoms_that_changed_rank": int(per_rank[:, 3].sum()),
            "note": "same %d steps; every %d steps atoms are wrapped and re-homed, ghosts re-derived (exchange + borders) and the "
                    "full neighbour list (cutoff %.3f A) rebuilt on the device" % (args.steps, args.rebuild_every, rc_list)},
    }
    if dry:
        out["rehearsal"] = "ANNP_BENCH_DRYRUN=1: control flow only (ranks on CPU over gloo, no force evaluation, nothing measured)"
        _RESULT_LINE.append(json.dumps(out))
        if use_dist:
            dist.destroy_process_group()
        return
    achieved = flop_force / (force_ms * 1e-3) / 1e12
    traffic, traffic_src = _pmc_traffic("annp_fe_force<" if pair_loop else "annp_fe_force_sh", natoms if world == 1 else None)
    out["kernel_ms"] = {"descriptor": desc_ms, "network": mlp_ms, "force": force_ms, "evaluation": float(ms4[3]), "samples": int(ns.value),
                        "rank": 0}
    out["roofline"] = {
            "kernel": "annp_fe_force<9,19> (pair loop)" if pair_loop else "annp_fe_force_sh<9,19>",
            "bound": "fp64_valu",
            "achieved": achieved,
            "peak": PEAK_FP64_VECTOR,
            "unit": "TFLOP/s",
            "frac": achieved / PEAK_FP64_VECTOR,
            "traffic": traffic, "traffic_source": traffic_src,
            "algorithmic_flop_per_launch": flop_force,
            "flop_model": "kernel-derived, DESIGN.md 4.5 (what the implemented formulation executes; SURVEY.md 8d's budget for the "
                          "reference formulation is survey_budget below)",
            "flop_per_unit": {"pair": fpf, "neighbour": fnf, "atom": faf},
            "descriptor_pass": {"achieved": flop_desc / (desc_ms * 1e-3) / 1e12, "frac": flop_desc / (desc_ms * 1e-3) / 1e12 / PEAK_FP64_VECTOR,
                                "flop_per_unit": {"pair": fpd, "neighbour": fnd, "atom": fad}},
            "whole_evaluation": {"achieved": flop_eval / (float(ms4[3]) * 1e-3) / 1e12,
                                 "frac": flop_eval / (float(ms4[3]) * 1e-3) / 1e12 / PEAK_FP64_VECTOR,
                                 "flop_per_atom_step": flop_eval / nlocal},
            "survey_budget": {"flop_per_atom_step": flop_survey / nlocal,
                              "equivalent_TFLOPs": flop_survey / (float(ms4[3]) * 1e-3) / 1e12,
                              "note": "SURVEY.md 8d prices the reference formulation (2.20 MFLOP per atom-step, O(n^2) pairs per atom); the "
                                      "kernels reach the same result from the moments of the neighbourhood with about a fifth of that, so "
                                      "this figure is not a pipe utilisation"},
            "hbm": {"achieved_GBps": nlocal * BYTES_ATOM_STEP / (float(ms4[3]) * 1e-3) / 1e9, "peak_GBps": PEAK_HBM,
                    "note": "%.1f KB per atom-step over the whole evaluation; the path is FP64-VALU bound, not HBM bound" % (BYTES_ATOM_STEP / 1e3)},
    }
    if wl == "fe" and not pair_loop:
        ex = _pmc_extras(natoms if world == 1 else None, "annp_fe_force_sh")
        for k in ("descriptor_pass", "network_pass"):
            if k in ex:
                out["roofline"].setdefault(k, {}).update(ex.pop(k))
        out["roofline"].update(ex)
        if "peak_at_sustained_clock" in out["roofline"]:
            out["roofline"]["frac_of_peak_at_sustained_clock"] = achieved / out["roofline"]["peak_at_sustained_clock"]
    if world > 1:
        out["roofline"]["note"] = "rank 0's launches (its %d owned atoms)" % nlocal
    if wl == "ni":
        ev = flop_eval / (float(ms4[3]) * 1e-3) / 1e12
        out["roofline"] = {
            "kernel": "annp_ni_desc + annp_mlp_mfma + annp_ni_force (whole evaluation)", "bound": "fp64_valu",
            "achieved": ev, "peak": PEAK_FP64_VECTOR, "unit": "TFLOP/s", "frac": ev / PEAK_FP64_VECTOR, "traffic": None,
            "algorithmic_flop_per_launch": flop_eval,
            "flop_per_unit": {"candidate_pair": NI_FLOP_PAIR, "neighbour": NI_FLOP_NBR, "atom": NI_FLOP_MLP},
            "survey_budget_TFLOPs": flop_survey / (float(ms4[3]) * 1e-3) / 1e12,
            "note": "flop counted from the kernels (DESIGN.md 4.5): ~50 kflop per atom-step, a third of SURVEY.md 8d's 0.17 MFLOP "
                    "estimate; with ~18 neighbours per atom the passes are bound by LDS and memory latency at 3-4 waves per SIMD, "
                    "not by FP64 issue and not by HBM (DESIGN.md 4.4)"}
    elif wl == "anna":
        dd = flop_desc / (desc_ms * 1e-3) / 1e12
        out["roofline"] = {
            "kernel": "annp_fe_desc_sh<9,19> (descriptor pass of pair_style anna_adp)", "bound": "fp64_valu",
            "achieved": dd, "peak": PEAK_FP64_VECTOR, "unit": "TFLOP/s", "frac": dd / PEAK_FP64_VECTOR, "traffic": None,
            "algorithmic_flop_per_launch": flop_desc, "flop_per_unit": {"pair": fpd, "neighbour": fnd, "atom": fad},
            "note": "the second kernel (network + ADP sums + forces, no pair loop) is bound by its scattered force atomics "
                    "and memory latency, not by arithmetic (DESIGN.md 4.4b)"}
    # ---- CPU baseline (rank 0, N = 1 only) ------------------------------------------------
    if world == 1 and args.cpu_sample > 0:
        # (the only leg that touches the oracle, through the tests' loader: after the timed regions, as the checker's clock)
        from annp_testlib import FAST, KIND_FE, KIND_NI_FIXED, anna_compute, oracle_compute, oracle_lib, read_anna, read_pot
        m = min(args.cpu_sample, nlocal)
        xs = dom.x.cpu().numpy()
        s = _sample_system(lib, h, xs, nlocal, nall, m, rc_list)
        nthreads = min(oracle_lib().annp_oracle_max_threads(), _cpu_share())
        if wl == "anna":
            C.CDLL("libgomp.so.1").omp_set_num_threads(int(nthreads))       # this oracle takes OpenMP's default
            pot = read_anna(ANNA_POT)
            anna_compute(pot, s, inum=min(m, 256))                                               # warm
            t1 = time.perf_counter()
            anna_compute(pot, s, inum=m)
        else:
            pot = read_pot(potfile)
            kind = KIND_FE if wl == "fe" else KIND_NI_FIXED
            oracle_compute(pot, s, kind, FAST, inum=min(m, 256), nthreads=nthreads)              # warm
            t1 = time.perf_counter()
            oracle_compute(pot, s, kind, FAST, inum=m, nthreads=nthreads)
        tc = time.perf_counter() - t1
        out["cpu_baseline"] = {
            "value": m / tc, "unit": "atom-steps/s", "cores": int(nthreads), "kind": "port",
            "sample": "1 force evaluation of the first %d of %d atoms of the same box (same neighbour list), "
                      "oracle FAST strategy, OpenMP" % (m, natoms),
            "seconds": tc,
            "note": "the literal reference CPU pair_annp measured during the survey: 129 atom-steps/s on one core at "
                    "2 000 atoms (BASELINE.md 2); it cannot run at this size (O(N nall) allocations)",
        }
        if wl == "fe":
            try:
                out["cpu_baseline"]["port_1024000"] = {k: out["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "sample", "seconds")}
                out["cpu_baseline"].update(_cpu_matrix(nthreads))
            except Exception as exc:    # the metric's line must not be lost to a baseline
                out["cpu_baseline"]["matrix_error"] = repr(exc)
    out["config"]["evaluations_reissued"] = leg.reissued
    out["config"]["spin_up"] = {"evaluations": spun, "note": "untimed force evaluations on the start configuration before the W warm-up steps, until the device "
                                                               "has been busy for ANNP_BENCH_SPINUP_MS (60) ms: the chip needs 20-30 ms of work after the host-bound "
                                                               "set-up to reach the clocks it then holds; no MD step, no work of the timed region is done there"}
    # ---- BASELINE.json config 5, and the reference's published deck, as short extra legs of the default run ----
    if world == 1 and wl == "fe" and args.secondary and not use_dist:
        leg.close()
        del leg, dom
        torch.cuda.empty_cache()
        try:
            out["secondary"] = {"ni": secondary_ni(args, dev, local_rank)}
        except Exception as exc:        # the metric's line must not be lost to the extra leg
            out["secondary"] = {"ni": {"error": repr(exc)}}
        torch.cuda.empty_cache()
        try:                            # the reference's own published deck
            out["secondary"]["fe_st"] = secondary_fe_st(args, dev, local_rank)
        except Exception as exc:
            out["secondary"]["fe_st"] = {"error": repr(exc)}
    _RESULT_LINE.append(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


def _pmc_profile(natoms, kernel=None, element=None):
    """the newest committed counter summary of this command at this size (profiles/r*_pmc_counters.json, written by
    tools/summarise_profiles.py from tools/collect_profiles.sh's passes) that has an entry for `kernel` -- a run with the
    pair-loop kernels finds the round that profiled them, not the newest file -- and names `element` in its workload;
    or (None, None)"""
    import glob
    best, src = None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_counters.json"))):
        try:
            d = json.load(open(f))
            w = d.get("workload", "")
            if natoms is None or str(natoms) not in w or "anna" in w or (element and element not in w):
                continue
            if kernel is not None and _pmc_kernel(d["per_launch_mean"], kernel) is None:
                continue
            best, src = d["per_launch_mean"], os.path.relpath(f, ROOT)
        except Exception:
            pass
    return best, src


def _pmc_kernel(prof, kernel):
    """the steady-state instantiation of `kernel` in a counter summary (the one with the most HBM traffic: the first evaluation
    may run another instantiation once)"""
    cand = [v for k, v in (prof or {}).items() if kernel in k and "fixup" not in k and "hbm_bytes_upper" in v]
    return max(cand, key=lambda v: v["hbm_bytes_upper"]) if cand else None


def _pmc_traffic(kernel, natoms):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/r*_pmc_counters.json: FETCH_SIZE and WRITE_SIZE in separate passes; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950, i.e. the upper bound).  None when no matching profile exists:
    the counters cannot be read from inside an un-profiled run.  Returns (bytes, where they came from)."""
    prof, f = _pmc_profile(natoms, kernel)
    e = _pmc_kernel(prof, kernel)
    if e is None:
        return None, None
    return e["hbm_bytes_upper"], "%s (builder's own rocprofv3 --pmc passes of this command on an MI355X, FETCH_SIZE and WRITE_SIZE in " \
                                 "separate passes, FETCH_SIZE doubled per the gfx950 note: an upper bound; NOT measured in this run)" % f


def _pmc_extras(natoms, force_kernel):
    """What north_star asks rocprof to show, from the same committed passes: the shader clock the chip held under the force pass
    (GRBM_GUI_ACTIVE / 8 XCDs / the kernel's mean duration in the kernel trace of the same command), the descriptor pass's HBM
    bytes per second, the network pass's FP64-MFMA rate and matrix-pipe busy share.  Empty when no profile of this size exists."""
    prof, f = _pmc_profile(natoms, force_kernel)
    out = {}
    e = _pmc_kernel(prof, force_kernel)
    if e and e.get("clock_GHz"):
        out["sustained_clock_GHz"] = e["clock_GHz"]
        out["peak_at_sustained_clock"] = PEAK_FP64_VECTOR * e["clock_GHz"] / 2.4
        if e.get("SQ_INSTS_VALU") and e.get("GRBM_GUI_ACTIVE"):
            out["valu_instructions_per_atom"] = e["SQ_INSTS_VALU"] / natoms
            out["valu_issue_share"] = e["SQ_INSTS_VALU"] * 4 / (1024 * e["GRBM_GUI_ACTIVE"] / 8)
    d = _pmc_kernel(prof, "annp_fe_desc_sh")
    if d and d.get("hbm_GBps_upper"):
        out["descriptor_pass"] = {"hbm_GBps": d["hbm_GBps_upper"], "hbm_GBps_lower": d["hbm_GBps_lower"], "hbm_frac_of_8TBps": d["hbm_GBps_upper"] / PEAK_HBM,
                                  "hbm_bytes_per_launch": d["hbm_bytes_upper"], "avg_ms": d.get("avg_ms"),
                                  "valu_instructions_per_atom": d.get("SQ_INSTS_VALU", 0.0) / natoms,
                                  "lds_bank_conflict_share": (d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]) if d.get("SQ_LDS_IDX_ACTIVE") else None}
    m = _pmc_kernel(prof, "annp_mlp_mfma")
    if m and m.get("mfma_TFLOPs"):
        out["network_pass"] = {"mfma_TFLOPs": m["mfma_TFLOPs"], "mfma_util": m["mfma_TFLOPs"] / PEAK_FP64_VECTOR,
                               "mfma_busy_share": m.get("mfma_busy_share"), "hbm_GBps": m.get("hbm_GBps_upper"), "avg_ms": m.get("avg_ms")}
    if out:
        out["source"] = "%s: rocprofv3 --pmc passes + kernel trace of this command (tools/collect_profiles.sh), NOT measured in this run" % f
    return out


def _cpu_share():
    """CPUs this process may actually use: affinity mask capped by the cgroup quota."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def _sample_system(lib, h, x_all, nlocal, nall, m, rc_list):
    """Host copy of positions + neighbour list rows of the first m atoms (for the oracle)."""
    from annp_testlib import System
    s = System.__new__(System)
    s.nlocal, s.nall, s.nghost = nlocal, nall, nall - nlocal
    s.x = np.ascontiguousarray(x_all)
    s.type = np.ones(nall, dtype=np.int32)
    # build the sample's rows with the harness builder on the host (same criterion r^2 <= rc^2)
    from annp_testlib import _dp, _ip, _lp, oracle_lib
    ol = oracle_lib()
    s.numneigh = np.zeros(nall, dtype=np.int32)
    tot = ol.harness_neigh(m, nall, _dp(s.x), rc_list, _ip(s.numneigh), None, None)
    s.first = np.zeros(nall + 1, dtype=np.int64)
    np.cumsum(s.numneigh, out=s.first[1:])
    s.neigh = np.empty(max(int(tot), 1), dtype=np.int32)
    ol.harness_neigh(m, nall, _dp(s.x), rc_list, _ip(s.numneigh), _lp(s.first), _ip(s.neigh))
    s.ilist = np.arange(m, dtype=np.int32)
    s.inum = m
    s.owner = np.zeros(s.nghost, dtype=np.int32)
    s.rc_list = rc_list
    return s


def _host_sample(x_local, box, m, rc_list):
    """A periodic box on the host, as the oracle wants it: ghosts from the test harness, neighbour-list rows of the first m atoms"""
    from annp_testlib import System, _dp, _ip, _lp, oracle_lib
    ol = oracle_lib()
    x_local = np.ascontiguousarray(x_local, dtype=np.float64)
    n = x_local.shape[0]
    boxa = np.ascontiguousarray(box, dtype=np.float64)
    per = np.ones(3, dtype=np.int32)
    ng = ol.harness_ghosts(n, _dp(x_local), _dp(boxa), _ip(per), rc_list, 0, None, None)
    xg = np.empty((ng, 3))
    owner = np.empty(ng, dtype=np.int32)
    ol.harness_ghosts(n, _dp(x_local), _dp(boxa), _ip(per), rc_list, ng, _dp(xg), _ip(owner))
    s = System.__new__(System)
    s.nlocal, s.nghost, s.nall = n, int(ng), n + int(ng)
    s.x = np.ascontiguousarray(np.vstack([x_local, xg]))
    s.type = np.ones(s.nall, dtype=np.int32)
    s.numneigh = np.zeros(s.nall, dtype=np.int32)
    tot = ol.harness_neigh(m, s.nall, _dp(s.x), rc_list, _ip(s.numneigh), None, None)
    s.first = np.zeros(s.nall + 1, dtype=np.int64)
    np.cumsum(s.numneigh, out=s.first[1:])
    s.neigh = np.empty(max(int(tot), 1), dtype=np.int32)
    ol.harness_neigh(m, s.nall, _dp(s.x), rc_list, _ip(s.numneigh), _lp(s.first), _ip(s.neigh))
    s.ilist = np.arange(m, dtype=np.int32)
    s.inum, s.owner, s.rc_list = m, owner, rc_list
    return s


def _cpu_matrix(nthreads):
    """SURVEY.md 8d's CPU figures beside the GPU's, each a bounded sample (a few seconds in all): the oracle's LITERAL strategy
    (one atom at a time, dG/dx materialised per list slot, forward-mode Jacobian, the reference's order of operations -- the closest
    thing to fe_v2/src/pair_annp.cpp:74-218 that can be timed where the reference itself cannot be built) on ONE core at 2 000
    atoms, and the allocation-free port (FAST, OpenMP) at 2 000 and 128 000 atoms."""
    from annp_testlib import FAST, KIND_FE, LITERAL, oracle_compute, read_pot
    from meng_zhang_amd.workloads import A_FE, FE_POT, bcc, perturb
    pot = read_pot(FE_POT)
    out = {}
    x, box = bcc(10, 10, 10, A_FE)
    s = _host_sample(perturb(x, 12345, 0.05), box, 2000, 8.5)
    oracle_compute(pot, s, KIND_FE, LITERAL, inum=16, nthreads=1)
    t = time.perf_counter()
    oracle_compute(pot, s, KIND_FE, LITERAL, inum=2000, nthreads=1)
    dt = time.perf_counter() - t
    out["literal_2000"] = {"value": 2000 / dt, "unit": "atom-steps/s", "cores": 1, "kind": "port (LITERAL strategy of the oracle)", "seconds": dt,
                           "sample": "one evaluation of the whole 2 000-atom box (BASELINE.json config 0's system) on one core",
                           "reference_carried_over": {"value": 129.0, "unit": "atom-steps/s per core",
                                                      "note": "the unmodified reference translation unit measured during the survey (8 vCPU Xeon @ 2.1 GHz, "
                                                              "BASELINE.md 2): several times slower than this restatement because it allocates and frees "
                                                              "(nall+2) x nsf small arrays per atom (fe/src/pair_annp.cpp:122-130); it cannot be built here"}}
    oracle_compute(pot, s, KIND_FE, FAST, inum=256, nthreads=nthreads)
    t = time.perf_counter()
    oracle_compute(pot, s, KIND_FE, FAST, inum=2000, nthreads=nthreads)
    dt = time.perf_counter() - t
    out["port_2000"] = {"value": 2000 / dt, "unit": "atom-steps/s", "cores": int(nthreads), "kind": "port", "seconds": dt,
                        "sample": "one evaluation of the whole 2 000-atom box, oracle FAST strategy, OpenMP"}
    x, box = bcc(40, 40, 40, A_FE)
    m = 32768
    s = _host_sample(perturb(x, 12345, 0.05), box, m, 8.5)
    t = time.perf_counter()
    oracle_compute(pot, s, KIND_FE, FAST, inum=m, nthreads=nthreads)
    dt = time.perf_counter() - t
    out["port_128000"] = {"value": m / dt, "unit": "atom-steps/s", "cores": int(nthreads), "kind": "port", "seconds": dt,
                          "sample": "one evaluation of the first %d atoms of the 128 000-atom box (BASELINE.json config 1's system), oracle FAST strategy, OpenMP" % m}
    return out


def _main_one_json_line():
    """stdout carries exactly one line, the JSON: whatever libraries print while the run is set up
    (e.g. RCCL's version banner under NCCL_DEBUG=VERSION) is sent to stderr instead."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        main()
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)
    if _RESULT_LINE:
        print(_RESULT_LINE[0], flush=True)


_RESULT_LINE = []

if __name__ == "__main__":
    _args = parse_args()
    if _args.gpus > 1 and "WORLD_SIZE" not in os.environ:       # not under a launcher: become one (no GPU call has happened)
        sys.exit(launch_ranks(_args))
    _main_one_json_line()
