"""Spatial decomposition of one periodic box over the GPUs of a node.

What LAMMPS core does around ``Pair::compute`` for the reference (``processors N 1 1``
+ ``Comm::forward_comm`` / ``reverse_comm`` with ``newton on``),
restated for one process per GPU with ``torch.distributed`` (backend ``nccl`` = RCCL over
xGMI on ROCm, ``gloo`` on CPU in the tests):

* the box is cut into ``world`` slabs along x; a rank owns the atoms whose home
  coordinate falls into its slab and carries *ghost* copies of every atom image that
  lies within ``rc_halo`` of the slab (periodic images in y and z, neighbours' atoms
  and periodic images in x);
* ``forward()``  -- owners send current positions, ghosts receive them (+ image shift);
* ``reverse()``  -- ghosts send the forces they accumulated back, owners add them.

Both directions are one group of point-to-point sends/receives between slab
neighbours (``batch_isend_irecv`` = ncclGroupStart/ncclSend/ncclRecv/ncclGroupEnd):
xGMI is point-to-point, a slab has two neighbours, so each exchange uses two links
and moves only the boundary atoms.  No collective touches the data path; only the
scalar energy (and virial) is all-reduced.

The plan (who sends what to whom) is computed once from the initial configuration,
identically on every rank, so no communication is needed to set it up.  Arithmetic is
not done here: the force engine is passed in (the HIP library in the product; the
tests may pass any callable with the same signature to check the decomposition).
"""
import numpy as np

_SHIFTS = [(sx, sy, sz) for sx in (-1, 0, 1) for sy in (-1, 0, 1) for sz in (-1, 0, 1)]


def slab_of(x_home, box, world):
    """Rank that owns each atom: equal-width slabs along x."""
    lx = box[3] - box[0]
    r = np.floor((x_home[:, 0] - box[0]) / lx * world).astype(np.int64)
    return np.clip(r, 0, world - 1)


def _halo_block(x_src, box, periodic, rc, lo, hi, owner_is_self):
    """Images of x_src (atoms of one owner rank) that fall into the halo of the slab
    [lo, hi) x box_y x box_z, in (shift, local index) order.
    Returns (local indices, shift vectors)."""
    L = np.array([box[3] - box[0], box[4] - box[1], box[5] - box[2]])
    idx_all, sh_all = [], []
    for s in _SHIFTS:
        if any(s[d] != 0 and not periodic[d] for d in range(3)):
            continue
        if owner_is_self and s == (0, 0, 0):
            continue
        p = x_src + np.array(s, dtype=np.float64) * L
        m = (p[:, 0] >= lo - rc) & (p[:, 0] < hi + rc)
        m &= (p[:, 1] >= box[1] - rc) & (p[:, 1] < box[4] + rc)
        m &= (p[:, 2] >= box[2] - rc) & (p[:, 2] < box[5] + rc)
        if not owner_is_self and s == (0, 0, 0):
            pass    # another rank's atoms inside my halo region
        # an image that lies inside my own slab (possible only for s != 0 of far atoms) is a real ghost too
        ids = np.nonzero(m)[0]
        if ids.size:
            idx_all.append(ids)
            sh_all.append(np.tile(np.array(s, dtype=np.float64) * L, (ids.size, 1)))
    if not idx_all:
        return np.zeros(0, dtype=np.int64), np.zeros((0, 3))
    return np.concatenate(idx_all), np.vstack(sh_all)


class HaloPlan:
    """Ownership and exchange lists for one rank (pure numpy; same on CPU and GPU runs)."""

    def __init__(self, x_global, box, periodic, rc_halo, world, rank):
        box = np.asarray(box, dtype=np.float64)
        self.world, self.rank, self.box, self.rc = world, rank, box, rc_halo
        owner = slab_of(x_global, box, world)
        lx = box[3] - box[0]
        if lx / world < rc_halo and world > 1:
            raise ValueError("slab thinner than the halo: %g < %g" % (lx / world, rc_halo))
        self.own_ids = [np.nonzero(owner == r)[0] for r in range(world)]       # global ids per rank, local order
        self.nlocal = int(self.own_ids[rank].size)

        def slab(r):
            return box[0] + lx * r / world, box[0] + lx * (r + 1) / world

        # ghosts of THIS rank, grouped by owner rank q
        lo, hi = slab(rank)
        self.recv_counts = np.zeros(world, dtype=np.int64)
        g_local, g_shift, g_owner = [], [], []
        for q in range(world):
            idx, sh = _halo_block(x_global[self.own_ids[q]], box, periodic, rc_halo, lo, hi, q == rank)
            self.recv_counts[q] = idx.size
            g_local.append(idx); g_shift.append(sh); g_owner.append(np.full(idx.size, q, dtype=np.int64))
        self.ghost_owner = np.concatenate(g_owner)
        self.ghost_owner_local = np.concatenate(g_local)       # index in the owner's local order
        self.ghost_shift = np.vstack(g_shift)
        self.nghost = int(self.ghost_owner.size)
        self.nall = self.nlocal + self.nghost
        # what THIS rank sends to every other rank r (= r's ghost block owned by me), in r's order
        self.send_idx = []
        for r in range(world):
            if r == rank:
                self.send_idx.append(self.ghost_owner_local[self.ghost_owner == rank])
                continue
            lo_r, hi_r = slab(r)
            idx, _ = _halo_block(x_global[self.own_ids[rank]], box, periodic, rc_halo, lo_r, hi_r, False)
            self.send_idx.append(idx)
        self.send_counts = np.array([s.size for s in self.send_idx], dtype=np.int64)
        self.recv_offsets = np.concatenate([[0], np.cumsum(self.recv_counts)])

    def local_positions(self, x_global):
        """[owned | ghosts] positions for this rank from a global configuration."""
        xo = x_global[self.own_ids[self.rank]]
        xg = np.empty((self.nghost, 3))
        for q in range(self.world):
            a, b = self.recv_offsets[q], self.recv_offsets[q + 1]
            xg[a:b] = x_global[self.own_ids[q]][self.ghost_owner_local[a:b]] + self.ghost_shift[a:b]
        return np.vstack([xo, xg])


class Domain:
    """Device (or CPU) state of one rank + the two halo exchanges."""

    def __init__(self, plan, x_global, device, dist=None):
        import torch
        self.torch, self.dist, self.plan, self.device = torch, dist, plan, device
        p = plan
        self.x = torch.from_numpy(p.local_positions(x_global)).to(device).contiguous()
        self.f = torch.zeros_like(self.x)
        self.shift = torch.from_numpy(p.ghost_shift).to(device)
        self.send_idx = [torch.from_numpy(s.astype(np.int64)).to(device) for s in p.send_idx]
        self.peers = [q for q in range(p.world) if q != p.rank and (p.send_counts[q] or p.recv_counts[q])]
        # staging buffers (contiguous per peer)
        self.sbuf = {q: torch.empty((int(p.send_counts[q]), 3), dtype=torch.float64, device=device) for q in self.peers}
        self.rbuf = {q: torch.empty((int(p.recv_counts[q]), 3), dtype=torch.float64, device=device) for q in self.peers}
        self.bytes_per_exchange = sum(int(p.send_counts[q]) for q in self.peers) * 24

    # ---- Comm::forward_comm: positions owners -> ghosts
    def forward(self):
        t, p = self.torch, self.plan
        n = p.nlocal
        xo = self.x[:n]
        ops = []
        for q in self.peers:
            if p.send_counts[q]:
                t.index_select(xo, 0, self.send_idx[q], out=self.sbuf[q])
                ops.append(self.dist.P2POp(self.dist.isend, self.sbuf[q], q))
            if p.recv_counts[q]:
                ops.append(self.dist.P2POp(self.dist.irecv, self.rbuf[q], q))
        reqs = self.dist.batch_isend_irecv(ops) if ops else []
        # images of my own atoms need no wire
        a, b = int(p.recv_offsets[p.rank]), int(p.recv_offsets[p.rank + 1])
        if b > a:
            self.x[n + a:n + b] = xo.index_select(0, self.send_idx[p.rank]) + self.shift[a:b]
        for r in reqs:
            r.wait()
        for q in self.peers:
            a, b = int(p.recv_offsets[q]), int(p.recv_offsets[q + 1])
            if b > a:
                self.x[n + a:n + b] = self.rbuf[q] + self.shift[a:b]

    # ---- Comm::reverse_comm: ghost forces -> owners (newton_pair on, fe_v2/src/pair_annp.cpp:199)
    def reverse(self):
        p = self.plan
        n = p.nlocal
        ops = []
        for q in self.peers:
            a, b = int(p.recv_offsets[q]), int(p.recv_offsets[q + 1])
            if b > a:     # what I received positions for, I return forces for
                self.rbuf[q].copy_(self.f[n + a:n + b])
                ops.append(self.dist.P2POp(self.dist.isend, self.rbuf[q], q))
            if p.send_counts[q]:
                ops.append(self.dist.P2POp(self.dist.irecv, self.sbuf[q], q))
        reqs = self.dist.batch_isend_irecv(ops) if ops else []
        fo = self.f[:n]
        a, b = int(p.recv_offsets[p.rank]), int(p.recv_offsets[p.rank + 1])
        if b > a:
            fo.index_add_(0, self.send_idx[p.rank], self.f[n + a:n + b])
        for r in reqs:
            r.wait()
        for q in self.peers:
            if p.send_counts[q]:
                fo.index_add_(0, self.send_idx[q], self.sbuf[q])
