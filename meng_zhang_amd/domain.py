"""Spatial decomposition of one periodic box over the GPUs of a node.

What LAMMPS core does around ``Pair::compute`` for the reference (``processors N 1 1``;
``Comm::exchange`` + ``Comm::borders`` at every reneighbouring, ``Comm::forward_comm`` /
``reverse_comm`` with ``newton on`` every step), restated for one process
per GPU with ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI on ROCm, ``gloo`` on CPU
in the tests):

* the box is cut into ``world`` slabs along x; a rank owns the atoms inside its slab and
  carries *ghost* copies of every atom image within ``rc_halo`` (= list cutoff) of it;
* ``replan()``   -- at every neighbour-list rebuild: atoms are wrapped into the box, atoms that
  left the slab migrate to the neighbouring rank (``exchange``), and the ghost lists are
  re-derived from the current positions (``borders``): first across the two slab faces over
  the wire (counts first, then the atoms), then the periodic images in y and z of everything
  held so far as local copies -- so edges and corners come out right without diagonal messages;
* ``forward()``  -- owners' current positions -> ghosts (+ image shift);
* ``reverse()``  -- forces accumulated on ghosts -> owners.

Both per-step directions are one group of point-to-point sends/receives between slab
neighbours (``batch_isend_irecv`` = ncclGroupStart/ncclSend/ncclRecv/ncclGroupEnd): xGMI is
point-to-point, a slab has two neighbours, so an exchange uses two links and moves only the
boundary atoms, received straight into the ghost rows of ``x``.  No collective touches the
data path; collectives carry a handful of integers at a replan (message sizes) and scalars
(energy, largest displacement).  Nothing in the per-step path waits on the host.

Arithmetic is not done here: the force engine is passed the arrays (the HIP library in the
product; the tests pass the CPU oracle to check the decomposition itself).
"""
import numpy as np


class NoTransport:
    """single rank: no peers"""
    world, rank = 1, 0

    def route(self, msgs):
        assert not msgs

    def allgather(self, t):
        return t.reshape(1, -1).cpu()

    def allreduce_max(self, v):
        return float(v)

    def allreduce_sum_(self, t):
        return t


class TorchTransport:
    """torch.distributed (nccl = RCCL on the GPUs, gloo on CPU)"""

    def __init__(self, dist):
        self.dist = dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()

    def route(self, msgs):
        """msgs: list of ('send' | 'recv', tensor, peer); one ncclGroup of point-to-point transfers"""
        d = self.dist
        ops = [d.P2POp(d.isend if kind == "send" else d.irecv, t, peer) for kind, t, peer in msgs]
        for r in (d.batch_isend_irecv(ops) if ops else []):
            r.wait()

    def allgather(self, t):
        out = [t.new_empty(t.shape) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return self.dist_stack(out)

    @staticmethod
    def dist_stack(ts):
        import torch
        return torch.stack(ts).cpu()

    def allreduce_max(self, v):
        self.dist.all_reduce(v, op=self.dist.ReduceOp.MAX)
        return float(v)

    def allreduce_sum_(self, t):
        self.dist.all_reduce(t)
        return t


class LibTransport:
    """The halo wire in the library: point-to-point groups are issued by libannp_hip.so itself (annp_hip_comm_route: ncclGroupStart
    / ncclSend / ncclRecv / ncclGroupEnd on the caller's compute stream, no hop through torch's communication stream).  The small
    collectives of a re-planning (message sizes) and the scalars stay with `collectives` (a TorchTransport).
    The RCCL communicator is created here: rank 0 makes the unique id, `collectives` broadcasts its 128 bytes."""

    def __init__(self, lib, handle, collectives, device):
        import ctypes as C
        import torch
        self.lib, self.h, self.col, self.device, self.C, self.torch = lib, handle, collectives, device, C, torch
        self.world, self.rank = collectives.world, collectives.rank
        buf = C.create_string_buffer(128)
        if self.rank == 0:
            self._check(lib.annp_hip_comm_unique_id(buf), "comm_unique_id")
        ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).to(device)
        collectives.dist.broadcast(ident, 0)
        self._check(lib.annp_hip_comm_init(handle, bytes(ident.cpu().numpy().tobytes()), self.world, self.rank), "comm_init")

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.lib.annp_hip_last_error(self.h if what != "comm_unique_id" else None).decode()))

    def route(self, msgs):
        C = self.C
        msgs = [m for m in msgs if m[1].numel()]
        n = len(msgs)
        if n == 0:
            return
        for _, t, _ in msgs:
            assert t.is_contiguous() and t.dtype == self.torch.float64
        kinds = (C.c_int * n)(*[1 if k == "send" else 0 for k, _, _ in msgs])
        ptrs = (C.c_void_p * n)(*[t.data_ptr() for _, t, _ in msgs])
        cnts = (C.c_longlong * n)(*[t.numel() for _, t, _ in msgs])
        peers = (C.c_int * n)(*[p for _, _, p in msgs])
        st = self.torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.annp_hip_comm_route(self.h, n, kinds, ptrs, cnts, peers, st), "comm_route")

    def allgather(self, t):
        return self.col.allgather(t)

    def allreduce_max(self, v):
        return self.col.allreduce_max(v)

    def allreduce_sum_(self, t):
        return self.col.allreduce_sum_(t)


def slab_bounds(box, world, rank):
    lx = box[3] - box[0]
    return box[0] + lx * rank / world, box[0] + lx * (rank + 1) / world


class SlabDomain:
    """Atoms of one rank: ``x`` / ``f`` are ``[nall, 3]`` tensors, owned atoms first, then the ghosts received from
    the slab neighbours, then the local periodic images.  ``ids`` (global atom id) and every tensor in ``extra``
    (e.g. velocities) belong to the owned atoms and migrate with them."""

    def __init__(self, box, periodic, rc_halo, device, transport=None, x_own=None, ids=None, extra=None, hip=None, wire_self=False):
        import torch
        self.torch = torch
        self.tp = transport if transport is not None else NoTransport()
        self.world, self.rank = self.tp.world, self.tp.rank
        self.box = np.asarray(box, dtype=np.float64)
        self.periodic = tuple(bool(p) for p in periodic)
        self.rc = float(rc_halo)
        self.device = device
        L = self.box[3:] - self.box[:3]
        if (self.world > 1 or wire_self) and L[0] / self.world < self.rc:
            raise ValueError("slab thinner than the halo: %g < %g" % (L[0] / self.world, self.rc))
        self.lo, self.hi = slab_bounds(self.box, self.world, self.rank)
        w = self.world
        self.left = (self.rank - 1) % w if (self.rank > 0 or self.periodic[0]) else None
        self.right = (self.rank + 1) % w if (self.rank < w - 1 or self.periodic[0]) else None
        if w == 1:
            # one rank: every periodic image is a local copy -- unless wire_self asks for the x images to travel through
            # the transport to this same rank (the wire path of a slab whose two neighbours are itself: lets a single GPU
            # run ncclSend / ncclRecv on device buffers)
            self.left = self.right = (0 if (wire_self and self.periodic[0]) else None)
        self.wired = self.left is not None or self.right is not None
        x_own = torch.as_tensor(x_own, dtype=torch.float64, device=device).reshape(-1, 3).contiguous()
        self.nlocal = int(x_own.shape[0])
        self.x = x_own.clone()
        self.f = torch.zeros_like(self.x)
        self.ids = (torch.arange(self.nlocal, dtype=torch.int64, device=device) if ids is None
                    else torch.as_tensor(ids, dtype=torch.int64, device=device).clone())
        self.extra = {k: torch.as_tensor(v, dtype=torch.float64, device=device).reshape(self.nlocal, -1).clone()
                      for k, v in (extra or {}).items()}
        self.nghost = self.nall = 0
        self.n_replans = 0
        self.migrated_last = 0
        self.hip = hip                  # (lib, handle): the per-step jobs run as libannp_hip.so kernels (else torch ops)
        self.replan()

    @classmethod
    def from_global(cls, x_global, box, periodic, rc_halo, device, transport=None, extra=None, hip=None, wire_self=False):
        """Every rank holds the same global configuration (synthetic inputs) and keeps the atoms of its slab."""
        tp = transport if transport is not None else NoTransport()
        box = np.asarray(box, dtype=np.float64)
        L = box[3:] - box[:3]
        xw = np.array(x_global, dtype=np.float64, copy=True)
        for d in range(3):
            if periodic[d]:
                xw[:, d] -= np.floor((xw[:, d] - box[d]) / L[d]) * L[d]
        owner = np.clip(np.floor((xw[:, 0] - box[0]) / L[0] * tp.world).astype(np.int64), 0, tp.world - 1)
        mine = np.nonzero(owner == tp.rank)[0]
        ex = {k: np.asarray(v)[mine] for k, v in (extra or {}).items()}
        return cls(box, periodic, rc_halo, device, tp, x_global[mine], mine, ex, hip, wire_self)

    # ------------------------------------------------------------------ wire helpers
    def _route(self, send, n_l, n_r, recv, m_r, m_l):
        """send rows [to-left | to-right] -> neighbours; recv rows [from-right | from-left] <- neighbours.
        With two ranks both faces meet the same peer: one message each way carries both parts."""
        msgs = []
        if self.left is not None and self.left == self.right:
            if n_l + n_r:
                msgs.append(("send", send[: n_l + n_r], self.left))
            if m_r + m_l:
                msgs.append(("recv", recv[: m_r + m_l], self.left))
        else:
            if self.left is not None and n_l:
                msgs.append(("send", send[:n_l], self.left))
            if self.right is not None and n_r:
                msgs.append(("send", send[n_l:n_l + n_r], self.right))
            if self.right is not None and m_r:
                msgs.append(("recv", recv[:m_r], self.right))
            if self.left is not None and m_l:
                msgs.append(("recv", recv[m_r:m_r + m_l], self.left))
        return msgs

    def _route_back(self, ghost_rows, m_r, m_l, back, n_l, n_r):
        """the mirror image: rows [from-right | from-left] go back where they came from, arriving as [to-left | to-right]"""
        msgs = []
        if self.left is not None and self.left == self.right:
            if m_r + m_l:
                msgs.append(("send", ghost_rows[: m_r + m_l], self.left))
            if n_l + n_r:
                msgs.append(("recv", back[: n_l + n_r], self.left))
        else:
            if self.right is not None and m_r:
                msgs.append(("send", ghost_rows[:m_r], self.right))
            if self.left is not None and m_l:
                msgs.append(("send", ghost_rows[m_r:m_r + m_l], self.left))
            if self.left is not None and n_l:
                msgs.append(("recv", back[:n_l], self.left))
            if self.right is not None and n_r:
                msgs.append(("recv", back[n_l:n_l + n_r], self.right))
        return msgs

    def _peer_counts(self, n_l, n_r):
        """what my neighbours will send me, given what everyone sends: (from right = its to-left, from left = its to-right)"""
        t = self.torch
        mine = t.tensor([n_l, n_r], dtype=t.int64, device=self.device)
        allc = self.tp.allgather(mine).numpy().reshape(self.world, 2)
        m_r = int(allc[self.right][0]) if self.right is not None else 0
        m_l = int(allc[self.left][1]) if self.left is not None else 0
        return m_r, m_l

    # ------------------------------------------------------------------ Comm::exchange
    def _exchange(self):
        t = self.torch
        n = self.nlocal
        x = self.x[:n]
        L = self.box[3:] - self.box[:3]
        for d in range(3):
            if self.periodic[d]:
                x[:, d] -= t.floor((x[:, d] - self.box[d]) / L[d]) * L[d]
        self.migrated_last = 0
        if self.world == 1:
            return
        dest = t.clamp(t.floor((x[:, 0] - self.box[0]) / L[0] * self.world).to(t.int64), 0, self.world - 1)
        rel = (dest - self.rank) % self.world
        stay = t.nonzero(rel == 0).flatten()
        if self.world == 2:         # one other rank: it is my right neighbour, my left one (open box), or both (periodic)
            other = t.nonzero(rel != 0).flatten()
            go_r = other if self.right is not None else stay[:0]
            go_l = other if (self.right is None and self.left is not None) else stay[:0]
        else:
            go_r = t.nonzero(rel == 1).flatten() if self.right is not None else stay[:0]
            go_l = t.nonzero(rel == self.world - 1).flatten() if self.left is not None else stay[:0]
        if int(stay.numel() + go_r.numel() + go_l.numel()) != n:
            raise RuntimeError("an atom moved further than the neighbouring slab between two rebuilds")
        cols = [x, self.ids.to(t.float64).reshape(n, 1)] + [self.extra[k] for k in sorted(self.extra)]
        pay = t.cat(cols, dim=1)
        width = pay.shape[1]
        n_l, n_r = int(go_l.numel()), int(go_r.numel())
        m_r, m_l = self._peer_counts(n_l, n_r)
        send = pay[t.cat([go_l, go_r])].contiguous()
        recv = pay.new_empty((m_r + m_l, width))
        self.tp.route(self._route(send, n_l, n_r, recv, m_r, m_l))
        new = t.cat([pay[stay], recv], dim=0)
        self.migrated_last = n_l + n_r
        self.nlocal = int(new.shape[0])
        self._own_x = new[:, :3].contiguous()
        self.ids = new[:, 3].round().to(t.int64)
        c = 4
        for k in sorted(self.extra):
            wk = self.extra[k].shape[1]
            self.extra[k] = new[:, c:c + wk].contiguous()
            c += wk
        self.x = self._own_x

    # ------------------------------------------------------------------ Comm::borders
    def _borders(self):
        t = self.torch
        dev = self.device
        n = self.nlocal
        xo = self.x[:n]
        L = self.box[3:] - self.box[:3]
        # (1) across the slab faces
        if self.wired:
            sel_l = t.nonzero(xo[:, 0] < self.lo + self.rc).flatten() if self.left is not None else t.zeros(0, dtype=t.int64, device=dev)
            sel_r = t.nonzero(xo[:, 0] >= self.hi - self.rc).flatten() if self.right is not None else t.zeros(0, dtype=t.int64, device=dev)
            n_l, n_r = int(sel_l.numel()), int(sel_r.numel())
            m_r, m_l = self._peer_counts(n_l, n_r)
            self.send_idx = t.cat([sel_l, sel_r])
            sh = t.zeros((n_l + n_r, 3), dtype=t.float64, device=dev)
            if self.rank == 0:                       # my left neighbour sits at the far end of the box
                sh[:n_l, 0] = L[0]
            if self.rank == self.world - 1:
                sh[n_l:, 0] = -L[0]
            self.send_shift = sh
        else:
            n_l = n_r = m_r = m_l = 0
            self.send_idx = t.zeros(0, dtype=t.int64, device=dev)
            self.send_shift = t.zeros((0, 3), dtype=t.float64, device=dev)
        self.n_l, self.n_r, self.m_r, self.m_l = n_l, n_r, m_r, m_l
        nxg = m_r + m_l
        prim = t.empty((n + nxg, 3), dtype=t.float64, device=dev)
        prim[:n] = xo
        self.sendbuf = t.empty((n_l + n_r, 3), dtype=t.float64, device=dev)
        if self.wired:
            t.index_select(xo, 0, self.send_idx, out=self.sendbuf)
            self.sendbuf += self.send_shift
            self.tp.route(self._route(self.sendbuf, n_l, n_r, prim[n:], m_r, m_l))
        # (2) periodic images, one dimension after the other, of everything held so far
        root = t.arange(n + nxg, dtype=t.int64, device=dev)
        shift = t.zeros((n + nxg, 3), dtype=t.float64, device=dev)
        pos = prim
        dims = (1, 2) if self.wired else (0, 1, 2)
        for d in dims:
            if not self.periodic[d]:
                continue
            new_root, new_shift, new_pos = [], [], []
            for mask, s in ((pos[:, d] < self.box[d] + self.rc, L[d]), (pos[:, d] >= self.box[3 + d] - self.rc, -L[d])):
                idx = t.nonzero(mask).flatten()
                if idx.numel() == 0:
                    continue
                sv = shift[idx].clone()
                sv[:, d] += s
                pv = pos[idx].clone()
                pv[:, d] += s
                new_root.append(root[idx]); new_shift.append(sv); new_pos.append(pv)
            if new_root:
                root = t.cat([root] + new_root)
                shift = t.cat([shift] + new_shift)
                pos = t.cat([pos] + new_pos)
        np0 = n + nxg
        self.img_root = root[np0:].contiguous()
        self.img_shift = shift[np0:].contiguous()
        self.nxg = nxg
        self.nimg = int(self.img_root.numel())
        self.nghost = nxg + self.nimg
        self.nall = n + self.nghost
        self.x = pos.contiguous()                    # [owned | wire ghosts | images], already current
        self.f = t.zeros_like(self.x)
        self.backbuf = t.empty((n_l + n_r, 3), dtype=t.float64, device=dev)
        self.x_plan = self.x[:n].clone()
        # message lists of the per-step exchanges (tensors are views into x / f: valid until the next replan)
        self._fwd = self._route(self.sendbuf, n_l, n_r, self.x[n:n + nxg], m_r, m_l)
        self._rev = self._route_back(self.f[n:n + nxg], m_r, m_l, self.backbuf, n_l, n_r)
        self.bytes_per_exchange = (n_l + n_r) * 24
        if self.hip is not None:
            self._plan_hip()

    # ------------------------------------------------------------------ the per-step jobs as HIP kernels
    def _segments(self, targets):
        """rows k of a source array are added to row targets[k] of f: group them by target, in a fixed order
        (annp_hip_reverse_fold's seg_dst / seg_start / perm, int32)"""
        t = self.torch
        order = t.argsort(targets, stable=True)
        dst, counts = t.unique_consecutive(targets[order], return_counts=True)
        start = t.zeros(dst.numel() + 1, dtype=t.int32, device=self.device)
        if dst.numel():
            start[1:] = t.cumsum(counts, 0).to(t.int32)
        return dst.to(t.int32).contiguous(), start, order.to(t.int32).contiguous()

    def _plan_hip(self):
        t = self.torch
        self.send_idx32 = self.send_idx.to(t.int32).contiguous()
        self.img_root32 = self.img_root.to(t.int32).contiguous()
        d, st, pm = self._segments(self.img_root)
        self._img_seg = (int(d.numel()), d, st, pm)
        d, st, pm = self._segments(self.send_idx)
        self._back_seg = (int(d.numel()), d, st, pm)

    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def _hip_check(self, rc, what):
        if rc != 0:
            lib, h = self.hip
            raise RuntimeError("%s failed (%d): %s" % (what, rc, lib.annp_hip_last_error(h).decode()))

    def replan(self, eng=None):
        """Comm::exchange + Comm::borders: call whenever the neighbour list is rebuilt; f comes back zeroed, and so does the energy
        word `eng` when given (Verlet::force_clear of the reneighbouring step).  With the library at hand (hip=(lib, handle),
        atoms on the GPU) both run as its kernels (annp_hip_replan_*: stable stream compactions, the same arrays as the torch
        restatement below bit for bit); the wire and the exchange of message sizes stay here."""
        if self.hip is not None and self.device.type == "cuda" and len(self.extra) <= 2:
            self._exchange_hip()
            self._borders_hip(eng)
        else:
            self._exchange()
            self._borders()
            if eng is not None:
                eng.zero_()
        self.n_replans += 1

    # ------------------------------------------------------------------ the same two jobs as library kernels
    def _box_args(self):
        import ctypes as C
        return (C.c_double * 6)(*[float(v) for v in self.box]), (C.c_int * 3)(*[1 if q else 0 for q in self.periodic])

    def _exchange_hip(self):
        import ctypes as C
        t = self.torch
        lib, h = self.hip
        n = self.nlocal
        keys = sorted(self.extra)
        ex = [self.extra[k] for k in keys] + [None, None]
        w = [int(e.shape[1]) if e is not None else 0 for e in ex[:2]]
        box6, per3 = self._box_args()
        self.migrated_last = 0
        ptr = lambda a: a.data_ptr() if a is not None else None
        if self.world == 1:
            self._hip_check(lib.annp_hip_replan_exchange(h, n, self.x.data_ptr(), None, None, 0, None, 0, box6, per3, 1, 0, 0, 0, None, None, None,
                                                         self._stream()), "replan_exchange")
            return
        width = 4 + w[0] + w[1]
        cap = n + max(1024, n // 8)
        keep = t.empty((cap, width), dtype=t.float64, device=self.device)
        send = t.empty((max(n, 1), width), dtype=t.float64, device=self.device)
        counts = (C.c_int * 3)()
        self._hip_check(lib.annp_hip_replan_exchange(h, n, self.x.data_ptr(), self.ids.data_ptr(), ptr(ex[0]), w[0], ptr(ex[1]), w[1], box6, per3,
                                                     self.world, self.rank, int(self.left is not None), int(self.right is not None),
                                                     keep.data_ptr(), send.data_ptr(), counts, self._stream()), "replan_exchange")
        ns, n_l, n_r = int(counts[0]), int(counts[1]), int(counts[2])
        m_r, m_l = self._peer_counts(n_l, n_r)
        m = ns + m_r + m_l
        if m > cap:
            big = t.empty((m, width), dtype=t.float64, device=self.device)
            big[:ns] = keep[:ns]
            keep = big
        self.tp.route(self._route(send, n_l, n_r, keep[ns:m], m_r, m_l))
        self.migrated_last = n_l + n_r
        self.nlocal = m
        self._xcap = m + max(4096, m // 2)
        xb = t.empty((self._xcap, 3), dtype=t.float64, device=self.device)
        self.ids = t.empty(m, dtype=t.int64, device=self.device)
        new_ex = [t.empty((m, wk), dtype=t.float64, device=self.device) if wk else None for wk in w]
        self._hip_check(lib.annp_hip_replan_unpack(h, m, keep.data_ptr(), w[0], w[1], xb.data_ptr(), self.ids.data_ptr(), ptr(new_ex[0]), ptr(new_ex[1]),
                                                   self._stream()), "replan_unpack")
        for k, e in zip(keys, new_ex):
            self.extra[k] = e
        self._xbuf = xb
        self.x = xb[:m]

    def _borders_hip(self, eng=None):
        import ctypes as C
        t = self.torch
        lib, h = self.hip
        dev = self.device
        n = self.nlocal
        L = self.box[3:] - self.box[:3]
        box6, per3 = self._box_args()
        i32 = dict(dtype=t.int32, device=dev)
        f64 = dict(dtype=t.float64, device=dev)
        # the position buffer has room behind the owned atoms for the ghosts: [owned | wire ghosts | images]
        xb = getattr(self, "_xbuf", None)
        if xb is None or xb.data_ptr() != self.x.data_ptr() or xb.shape[0] < n:
            self._xcap = n + max(4096, n // 2)
            xb = t.empty((self._xcap, 3), **f64)
            xb[:n] = self.x[:n]
            self._xbuf = xb
        # (1) across the slab faces
        n_l = n_r = m_r = m_l = 0
        if self.wired:
            idx = t.empty(2 * max(n, 1), **i32)
            c2 = (C.c_int * 2)()
            self._hip_check(lib.annp_hip_replan_faces(h, n, xb.data_ptr(), float(self.lo + self.rc), float(self.hi - self.rc), int(self.left is not None),
                                                      int(self.right is not None), idx.data_ptr(), c2, self._stream()), "replan_faces")
            n_l, n_r = int(c2[0]), int(c2[1])
            m_r, m_l = self._peer_counts(n_l, n_r)
            self.send_idx32 = idx[: n_l + n_r]
            sh = t.zeros((n_l + n_r, 3), **f64)
            if self.rank == 0:                       # my left neighbour sits at the far end of the box
                sh[:n_l, 0] = L[0]
            if self.rank == self.world - 1:
                sh[n_l:, 0] = -L[0]
            self.send_shift = sh
        else:
            self.send_idx32 = t.empty(0, **i32)
            self.send_shift = t.empty((0, 3), **f64)
        self.send_idx = None                         # (the torch path's int64 copy: not needed here)
        self.n_l, self.n_r, self.m_r, self.m_l = n_l, n_r, m_r, m_l
        nxg = m_r + m_l
        np0 = n + nxg
        need = np0 + max(getattr(self, "nimg", 0) * 9 // 8, 1024)
        if xb.shape[0] < need:
            nb_ = t.empty((need + need // 4, 3), **f64)
            nb_[:n] = xb[:n]
            xb = self._xbuf = nb_
        self.sendbuf = t.empty((n_l + n_r, 3), **f64)
        if self.wired:
            if n_l + n_r:
                self._hip_check(lib.annp_hip_halo_pack(h, n_l + n_r, self.send_idx32.data_ptr(), self.send_shift.data_ptr(), xb.data_ptr(),
                                                       self.sendbuf.data_ptr(), self._stream()), "halo_pack")
            self.tp.route(self._route(self.sendbuf, n_l, n_r, xb[n:np0], m_r, m_l))
        # (2) periodic images of everything held so far (y and z; x too when nothing travels over the wire)
        dims_mask = 0b110 if self.wired else 0b111
        nimg = C.c_int(0)
        while True:
            cap = int(xb.shape[0])
            root = t.empty(max(cap - np0, 1), **i32)
            shift = t.empty((max(cap - np0, 1), 3), **f64)
            rc = lib.annp_hip_replan_images(h, np0, xb.data_ptr(), cap, box6, per3, float(self.rc), dims_mask, root.data_ptr(), shift.data_ptr(),
                                            C.byref(nimg), self._stream())
            if rc != -7:
                self._hip_check(rc, "replan_images")
                break
            nb_ = t.empty((np0 + int(nimg.value) + 4096, 3), **f64)      # not enough room: come back with more (rows < np0 are intact)
            nb_[:np0] = xb[:np0]
            xb = self._xbuf = nb_
        self.nxg, self.nimg = nxg, int(nimg.value)
        self.img_root32 = root[: self.nimg]
        self.img_shift = shift[: self.nimg]
        self.img_root = None
        self.nghost = nxg + self.nimg
        self.nall = n + self.nghost
        self.x = xb[: self.nall]
        if getattr(self, "_fbuf", None) is None or self._fbuf.shape[0] < self.nall:
            self._fbuf = t.empty((int(xb.shape[0]), 3), **f64)
        self.f = self._fbuf[: self.nall]
        self._hip_check(lib.annp_hip_halo_unpack_images(h, 0, None, None, None, 0, self.f.data_ptr(), 3 * self.nall,
                                                        eng.data_ptr() if eng is not None else None, self._stream()), "force clear")
        self.backbuf = t.empty((n_l + n_r, 3), **f64)
        self.x_plan = t.empty((n, 3), **f64)
        self.x_plan.copy_(self.x[:n])
        self._fwd = self._route(self.sendbuf, n_l, n_r, self.x[n:np0], m_r, m_l)
        self._rev = self._route_back(self.f[n:np0], m_r, m_l, self.backbuf, n_l, n_r)
        self.bytes_per_exchange = (n_l + n_r) * 24
        # the plans of the two folds: image forces onto their roots, returned rows onto the boundary atoms
        st_i, pm_i = t.empty(np0 + 1, **i32), t.empty(max(self.nimg, 1), **i32)
        self._hip_check(lib.annp_hip_replan_fold_plan(h, self.nimg, self.img_root32.data_ptr() if self.nimg else None, np0, st_i.data_ptr(),
                                                      pm_i.data_ptr(), self._stream()), "replan_fold_plan")
        self._img_seg = (np0, None, st_i, pm_i)
        st_b, pm_b = t.empty(n + 1, **i32), t.empty(max(n_l + n_r, 1), **i32)
        self._hip_check(lib.annp_hip_replan_fold_plan(h, n_l + n_r, self.send_idx32.data_ptr() if n_l + n_r else None, n, st_b.data_ptr(),
                                                      pm_b.data_ptr(), self._stream()), "replan_fold_plan")
        self._back_seg = (n, None, st_b, pm_b)

    # ------------------------------------------------------------------ per step
    def forward(self, clear_forces=False, eng=None):
        """Comm::forward_comm: positions owners -> ghosts.  clear_forces: also zero f (and the energy word `eng`) for the
        evaluation that follows -- Verlet::force_clear, fused into the image-fill launch on the HIP path."""
        t = self.torch
        n = self.nlocal
        np0 = n + self.nxg
        if self.hip is not None:
            lib, h = self.hip
            st = self._stream()
            if self._fwd:
                self._hip_check(lib.annp_hip_halo_pack(h, self.n_l + self.n_r, self.send_idx32.data_ptr(), self.send_shift.data_ptr(),
                                                       self.x.data_ptr(), self.sendbuf.data_ptr(), st), "halo_pack")
                self.tp.route(self._fwd)
            if self.nimg or clear_forces:
                self._hip_check(lib.annp_hip_halo_unpack_images(
                    h, self.nimg, self.img_root32.data_ptr(), self.img_shift.data_ptr(), self.x.data_ptr(), np0,
                    self.f.data_ptr() if clear_forces else None, 3 * self.nall if clear_forces else 0,
                    eng.data_ptr() if (clear_forces and eng is not None) else None, self._stream()), "halo_unpack_images")
            return
        if self._fwd:
            t.index_select(self.x[:n], 0, self.send_idx, out=self.sendbuf)
            self.sendbuf += self.send_shift
            self.tp.route(self._fwd)
        if self.nimg:
            t.index_select(self.x[:np0], 0, self.img_root, out=self.x[np0:])      # roots are owned atoms or wire ghosts
            self.x[np0:] += self.img_shift
        if clear_forces:
            self.f.zero_()
            if eng is not None:
                eng.zero_()

    def reverse(self):
        """Comm::reverse_comm: ghost forces -> owners (newton_pair on, fe_v2/src/pair_annp.cpp:199)"""
        n = self.nlocal
        np0 = n + self.nxg
        if self.hip is not None:
            lib, h = self.hip
            if self.nimg:       # images first: their roots may be wire ghosts, whose total then travels
                nseg, dst, start, perm = self._img_seg
                self._hip_check(lib.annp_hip_reverse_fold(h, nseg, dst.data_ptr() if dst is not None else None, start.data_ptr(), perm.data_ptr(),
                                                          self.f.data_ptr() + 24 * np0, self.f.data_ptr(), self._stream()), "reverse_fold")
            if self._rev:
                self.tp.route(self._rev)
                nseg, dst, start, perm = self._back_seg
                self._hip_check(lib.annp_hip_reverse_fold(h, nseg, dst.data_ptr() if dst is not None else None, start.data_ptr(), perm.data_ptr(),
                                                          self.backbuf.data_ptr(), self.f.data_ptr(), self._stream()), "reverse_fold")
            return
        if self.nimg:
            self.f[:np0].index_add_(0, self.img_root, self.f[np0:])
        if self._rev:
            self.tp.route(self._rev)
            self.f.index_add_(0, self.send_idx, self.backbuf)

    def verlet_half(self, v, dtf, dt=0.0):
        """FixNVE: v += dtf f, then x += dt v when dt != 0 (initial_integrate); dt = 0 is final_integrate"""
        n = self.nlocal
        if self.hip is not None:
            lib, h = self.hip
            self._hip_check(lib.annp_hip_verlet_half(h, n, self.x.data_ptr(), v.data_ptr(), self.f.data_ptr(), float(dtf), float(dt),
                                                     self._stream()), "verlet_half")
            return
        v.add_(self.f[:n], alpha=dtf)
        if dt != 0.0:
            self.x[:n].add_(v, alpha=dt)

    def max_displacement(self):
        """largest distance an owned atom has moved since the last replan, over all ranks
        (LAMMPS `neigh_modify check yes`: rebuild when it exceeds half the skin)"""
        t = self.torch
        d2 = ((self.x[: self.nlocal] - self.x_plan) ** 2).sum(1).max() if self.nlocal else t.zeros((), dtype=t.float64, device=self.device)
        return float(np.sqrt(self.tp.allreduce_max(d2.reshape(1).clone())))

    # ------------------------------------------------------------------ helpers for drivers / tests
    def gather_owned(self, values):
        """(ids, rows) of a per-owned-atom tensor as numpy, for assembling global arrays in tests"""
        return self.ids.cpu().numpy(), values[: self.nlocal].detach().cpu().numpy()
