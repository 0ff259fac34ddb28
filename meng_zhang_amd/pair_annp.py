"""Python face of the host-side pair style (host/annp_pair.h).

``PairANNP`` is used the way LAMMPS drives the reference class
(annp-gpu-lammps/fe_v2/src/pair_annp.h:24-32): ``settings`` -> ``coeff`` ->
``init_style`` -> ``init_one`` -> ``compute(eflag, vflag)``, with the LAMMPS objects the
reference reaches through ``atom->`` / ``list->`` passed as :class:`AtomData` /
:class:`NeighList`.  Errors the reference raises through ``error->all`` surface as
``RuntimeError`` with the same text.  All arithmetic happens in libannp_hip.so.
"""
import ctypes as C

import numpy as np

from .lib import load_library


class AtomData:
    """atom->x, atom->f, atom->type, nlocal, nghost (owned atoms first, ghosts after)."""

    def __init__(self, x, nlocal, type=None):
        self.x = np.ascontiguousarray(x, dtype=np.float64)
        self.nall = self.x.shape[0]
        self.nlocal = int(nlocal)
        self.nghost = self.nall - self.nlocal
        self.type = np.ones(self.nall, dtype=np.int32) if type is None else np.ascontiguousarray(type, dtype=np.int32)
        self.f = np.zeros_like(self.x)


class NeighList:
    """LAMMPS NeighList (full): ilist, numneigh[i], firstneigh[i] as CSR over ``neigh``."""

    def __init__(self, ilist, numneigh, first, neigh):
        self.ilist = np.ascontiguousarray(ilist, dtype=np.int32)
        self.inum = self.ilist.shape[0]
        self.numneigh = np.ascontiguousarray(numneigh, dtype=np.int32)
        self.first = np.ascontiguousarray(first, dtype=np.int64)
        self.neigh = np.ascontiguousarray(neigh, dtype=np.int32)
        # firstneigh: one pointer per atom into the flat array, as LAMMPS pages give
        base = self.neigh.ctypes.data
        n = self.numneigh.shape[0]
        self._ptrs = (C.POINTER(C.c_int) * n)()
        addr = base + self.first[:n] * 4
        for i in range(n):
            self._ptrs[i] = C.cast(int(addr[i]), C.POINTER(C.c_int))


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


class PairANNP:
    def __init__(self, ntypes=1, device=0, newton_pair=1, style="annp"):
        """style: "annp" (pair_style annp, .ann files) or "anna_adp" (pair_style anna_adp, .anna files)"""
        self._lib = load_library()
        self._p = self._lib.annp_pair_create_style(ntypes, style.encode())
        if not self._p:
            raise ValueError("unknown pair style %r" % style)
        self.style = style
        self.ntypes = ntypes
        self.device = device
        self.newton_pair = newton_pair
        self.eng_vdwl = 0.0
        self.eatom = None
        self.virial = np.zeros(6)
        self.vatom = None
        self.atom = None
        self.list = None
        self.ago = 0

    def close(self):
        if getattr(self, "_p", None):
            self._lib.annp_pair_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("%s (code %d)" % (self._lib.annp_pair_error(self._p).decode(), rc))

    @staticmethod
    def _argv(args):
        arr = (C.c_char_p * max(len(args), 1))()
        for i, a in enumerate(args):
            arr[i] = a.encode()
        return arr

    # ---- the reference's virtuals ------------------------------------------------
    def settings(self, args=()):
        self._check(self._lib.annp_pair_settings(self._p, len(args), self._argv(list(args))))

    def coeff(self, args):
        self._check(self._lib.annp_pair_coeff(self._p, len(args), self._argv(list(args))))

    def set_ni_compat(self, on):
        self._lib.annp_pair_set_ni_compat(self._p, int(bool(on)))

    def set_blocks_by_name(self, on):
        """potential files with several elements: False (default) = the reference parser's behaviour (every weight
        block lands in element 0), True = "#El" lines select the element of the blocks below them; before coeff()"""
        self._lib.annp_pair_set_blocks_by_name(self._p, int(bool(on)))

    def init_style(self):
        self._check(self._lib.annp_pair_init_style(self._p, self.newton_pair, self.device))

    def init_one(self, i, j):
        c = self._lib.annp_pair_init_one(self._p, i, j)
        if c < 0:
            self._check(-1)
        return c

    def compute(self, eflag=1, vflag=0, eflag_atom=True, vflag_atom=False):
        """PairANNP::compute(eflag, vflag) on self.atom / self.list; accumulates into atom.f."""
        a, l = self.atom, self.list
        if eflag_atom and (self.eatom is None or self.eatom.shape[0] != a.nall):
            self.eatom = np.zeros(a.nall)
        if vflag_atom and (self.vatom is None or self.vatom.shape[0] != a.nall):
            self.vatom = np.zeros((a.nall, 6))
        eng = C.c_double(0.0)
        vir = np.zeros(6)
        rc = self._lib.annp_pair_compute(self._p, int(eflag), int(vflag), int(bool(eflag_atom)), self.ago, l.inum, a.nall,
                                         a.nghost, _dp(a.x), _ip(a.type), _ip(l.ilist), _ip(l.numneigh), l._ptrs,
                                         _dp(a.f), C.byref(eng), _dp(self.eatom) if eflag_atom else None, _dp(vir),
                                         _dp(self.vatom) if vflag_atom else None)
        self._check(rc)
        self.ago += 1
        if eflag:
            self.eng_vdwl = eng.value          # pair_annp_gpu.cpp:127
        self.virial = vir
        return self.eng_vdwl

    def compute_n(self, cutneigh, eflag=1, vflag=0, eflag_atom=True, sublo=None, subhi=None):
        """Same, neighbour list built on the device (annp_gpu_compute_n analogue).  sublo/subhi are the
        sub-domain bounds LAMMPS would pass (domain->sublo/subhi); the library bins by the bounding box of the
        positions it is given, so they are optional here."""
        a = self.atom
        if eflag_atom and (self.eatom is None or self.eatom.shape[0] != a.nall):
            self.eatom = np.zeros(a.nall)
        eng = C.c_double(0.0)
        vir = np.zeros(6)
        lo = np.zeros(3) if sublo is None else np.ascontiguousarray(sublo, dtype=np.float64)
        hi = np.zeros(3) if subhi is None else np.ascontiguousarray(subhi, dtype=np.float64)
        rc = self._lib.annp_pair_compute_n(self._p, int(eflag), int(vflag), int(bool(eflag_atom)), self.ago, a.nlocal,
                                           a.nall, a.nghost, _dp(a.x), _ip(a.type), _dp(lo), _dp(hi), float(cutneigh),
                                           _dp(a.f), C.byref(eng), _dp(self.eatom) if eflag_atom else None, _dp(vir), None)
        self._check(rc)
        self.ago += 1
        if eflag:
            self.eng_vdwl = eng.value
        self.virial = vir
        return self.eng_vdwl

    def memory_usage(self):
        return self._lib.annp_pair_memory_usage(self._p)

    # ---- access for tests / drivers -----------------------------------------------
    @property
    def handle(self):
        return self._lib.annp_pair_handle(self._p)

    def potential(self):
        dims = np.zeros(8, dtype=np.int32)
        scal = np.zeros(5)
        act = np.zeros(8, dtype=np.int32)
        na, nb = np.zeros(64), np.zeros(64)
        self._check(self._lib.annp_pair_potential_info(self._p, _ip(dims), _dp(scal), _ip(act), _dp(na), _dp(nb)))
        ntl, nhl, nnod, nsf, npsf, ntsf, flagsym, has_sym = [int(v) for v in dims]
        out = dict(ntl=ntl, nhl=nhl, nnod=nnod, nsf=nsf, npsf=npsf, ntsf=ntsf, flagsym=flagsym, has_symcoef=has_sym,
                   cut=scal[0], e_scale=scal[1], e_shift=scal[2], e_atom=scal[3], mass=scal[4],
                   flagact=act[: ntl - 1].copy(), norm_a=na[:nsf].copy(), norm_b=nb[:nsf].copy(), W=[], B=[])
        nout = 1
        if self.style == "anna_adp":
            no, eb, es, gp = C.c_int(0), C.c_double(0), C.c_double(0), np.zeros(64)
            ngp = self._lib.annp_pair_potential_anna(self._p, C.byref(no), C.byref(eb), C.byref(es), _dp(gp), 64)
            if ngp < 0:
                self._check(ngp)
            nout = no.value
            out.update(nout=nout, e_base=eb.value, e_scal=es.value, gparams=gp[:ngp].copy())
        e = 0
        while True:                         # element e, layer l sits at index e * (ntl-1) + l; W / B are element 0's
            Ws, Bs = [], []
            for l in range(ntl - 1):
                nr = nout if l == ntl - 2 else nnod
                nc = nsf if l == 0 else nnod
                w, b = np.zeros(nr * nc), np.zeros(nr)
                if self._lib.annp_pair_potential_layer(self._p, e * (ntl - 1) + l, _dp(w), _dp(b)) != 0:
                    break
                Ws.append(w.reshape(nr, nc))
                Bs.append(b)
            if len(Ws) != ntl - 1:
                break
            if e == 0:
                out["W"], out["B"] = Ws, Bs
            out.setdefault("W_elem", []).append(Ws)
            out.setdefault("B_elem", []).append(Bs)
            e += 1
        if has_sym:
            rad, ang = np.zeros(npsf * 3), np.zeros(ntsf * 4)
            self._check(self._lib.annp_pair_potential_sym(self._p, _dp(rad), _dp(ang)))
            out["sym_rad"], out["sym_ang"] = rad.reshape(npsf, 3), ang.reshape(ntsf, 4)
        return out
