"""Test infrastructure for the N>1 path: a gloo transport (host buffers) and a slab-decomposed CPU
restatement of the V-cycle that mirrors, step for step, what libmg_hip.so does per rank
(slab geometry, halo exchanges, replicated coarse levels).  torch.distributed is used for moving bytes
only."""
import socket

import numpy as np


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def init_gloo(rank, world, port):
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    return dist


class GlooTransport:
    """The three callbacks of mg_set_comm_callbacks, on NumPy buffers."""

    def __init__(self, dist, rank, world):
        self.dist, self.rank, self.world = dist, rank, world

    def exchange(self, send_lo, send_hi, recv_lo, recv_hi):
        import torch
        reqs = []
        if recv_lo is not None:
            reqs.append(self.dist.irecv(torch.from_numpy(recv_lo), src=self.rank - 1))
        if recv_hi is not None:
            reqs.append(self.dist.irecv(torch.from_numpy(recv_hi), src=self.rank + 1))
        if send_lo is not None:
            reqs.append(self.dist.isend(torch.from_numpy(np.ascontiguousarray(send_lo)), dst=self.rank - 1))
        if send_hi is not None:
            reqs.append(self.dist.isend(torch.from_numpy(np.ascontiguousarray(send_hi)), dst=self.rank + 1))
        for r in reqs:
            r.wait()

    def allreduce(self, buf):
        import torch
        self.dist.all_reduce(torch.from_numpy(buf))

    def allgatherv(self, send, recv, counts):
        import torch
        off = 0
        for r in range(self.world):
            seg = recv[off:off + int(counts[r])]
            if r == self.rank:
                seg[:] = send
            self.dist.broadcast(torch.from_numpy(seg), src=r)
            off += int(counts[r])


def slab_splits(N0, level, world):
    """Plane boundaries of level `level` (N = N0 * 2**level): mirrors setup_geometry() in mg_capi.hip."""
    s = [((r * N0) // world) << level for r in range(world)]
    s.append(N0 * 2 ** level + 1)
    return s


class SlabOracle:
    """One rank's share of the V-cycle in NumPy/SciPy: same decomposition as the HIP library.

    Works in lexicographic numbering (hierarchies built with seed=None).  Levels with fewer than
    `replicate_below` unknowns (and always the coarsest) are replicated on every rank."""

    def __init__(self, bag, dim, transport, rank, world, replicate_below):
        self.bag, self.dim, self.t, self.rank, self.world = bag, dim, transport, rank, world
        self.lo, self.hi = bag.coarsest_level, bag.finest_level
        self.c = bag.coarsest_level_elements_per_dim
        self.N0 = self.c * 2 ** self.lo
        self.lv = {}
        for l in range(self.lo, self.hi + 1):
            N = self.c * 2 ** l
            n1 = N + 1
            plane = n1 ** (dim - 1)
            n = n1 ** dim
            rep = (l == self.lo) or n < replicate_below or world == 1
            sp_l = slab_splits(self.N0, l - self.lo, world)
            k0, k1 = (0, n1) if rep else (sp_l[rank], sp_l[rank + 1])
            h_lo = plane if (not rep and rank > 0) else 0
            h_hi = plane if (not rep and rank + 1 < world) else 0
            row0, nloc = k0 * plane, (k1 - k0) * plane
            A = bag.A_sp_dict[l][0].tocsr()
            rows = A[row0:row0 + nloc, :]
            lo_col = row0 - h_lo
            local = rows[:, lo_col:row0 + nloc + h_hi].tocsr()
            assert local.nnz == rows.nnz, "stencil reaches beyond one halo plane"
            d = A.diagonal()[row0:row0 + nloc]
            self.lv[l] = dict(N=N, n1=n1, plane=plane, rep=rep, k0=k0, k1=k1, h_lo=h_lo, h_hi=h_hi, row0=row0,
                              nloc=nloc, A=local, dinv=1.0 / d, splits=sp_l, Afull=A if rep else None)

    # local vector = [halo_lo | owned | halo_hi]
    def new(self, l):
        L = self.lv[l]
        return np.zeros(L["h_lo"] + L["nloc"] + L["h_hi"])

    def owned(self, l, x):
        L = self.lv[l]
        return x[L["h_lo"]:L["h_lo"] + L["nloc"]]

    def scatter(self, l, full):
        L = self.lv[l]
        x = self.new(l)
        x[:] = np.asarray(full).ravel()[L["row0"] - L["h_lo"]:L["row0"] + L["nloc"] + L["h_hi"]]
        return x

    def halo(self, l, x):
        L = self.lv[l]
        if L["rep"] or self.world == 1:
            return
        p, o = L["plane"], self.owned(l, x)
        self.t.exchange(o[:p].copy() if L["h_lo"] else None, o[-p:].copy() if L["h_hi"] else None,
                        x[:p] if L["h_lo"] else None, x[-p:] if L["h_hi"] else None)

    def smooth(self, l, v, f, nw):
        L = self.lv[l]
        w = self.bag.omega
        for _ in range(nw):
            o = self.owned(l, v)
            new = self.new(l)
            self.owned(l, new)[:] = o + w * L["dinv"] * (self.owned(l, f) - L["A"].dot(v))
            self.halo(l, new)
            v = new
        return v

    def grid(self, l, x):
        """View of [halo | owned | halo] as planes."""
        L = self.lv[l]
        return x.reshape((-1,) + (L["n1"],) * (self.dim - 1))

    def vcycle(self, l, v, f):
        from scipy.sparse.linalg import spsolve
        L = self.lv[l]
        if l == self.lo:
            out = self.new(l)
            out[:] = spsolve(L["Afull"], self.owned(l, f))
            return out
        C = self.lv[l - 1]
        v = self.smooth(l, v, f, self.bag.mu1)
        r = self.new(l)
        self.owned(l, r)[:] = self.owned(l, f) - L["A"].dot(v)
        # injection: coarse plane K <- fine plane 2K (same rank by construction)
        Fg = self.grid(l, self.owned(l, r))
        take = (slice(None, None, 2),) * (self.dim - 1)
        if C["rep"] and not L["rep"]:
            kc0, kc1 = C["splits"][self.rank], C["splits"][self.rank + 1]
        else:
            kc0, kc1 = C["k0"], C["k1"]
        mine = np.stack([Fg[2 * K - L["k0"]][take] for K in range(kc0, kc1)]).ravel()
        fc = self.new(l - 1)
        if C["rep"] and not L["rep"]:
            counts = np.array([(C["splits"][r + 1] - C["splits"][r]) * C["plane"] for r in range(self.world)])
            self.t.allgatherv(mine, fc, counts)
        else:
            self.owned(l - 1, fc)[:] = mine
        vc = self.vcycle(l - 1, self.new(l - 1), fc)
        # Q1 prolongation of the planes this rank owns
        self.halo(l - 1, vc)
        Cg = self.grid(l - 1, vc)                       # planes [kc_base, ...) incl. halos
        kc_base = C["k0"] - (1 if C["h_lo"] else 0)
        e = np.zeros((L["k1"] - L["k0"],) + (L["n1"],) * (self.dim - 1))
        for k in range(L["k0"], L["k1"]):
            kc = (k >> 1) - kc_base
            planes = [Cg[kc]] if k % 2 == 0 else [Cg[kc], Cg[kc + 1]]
            e[k - L["k0"]] = _interp_plane(planes, self.dim)
        out = v.copy()
        self.owned(l, out)[:] += e.ravel()
        self.halo(l, out)
        return self.smooth(l, out, f, self.bag.mu2)

    def gather(self, l, x):
        L = self.lv[l]
        if L["rep"] or self.world == 1:
            return self.owned(l, x).copy()
        full = np.zeros(L["n1"] ** self.dim)
        counts = np.array([(L["splits"][r + 1] - L["splits"][r]) * L["plane"] for r in range(self.world)])
        self.t.allgatherv(self.owned(l, x).copy(), full, counts)
        return full


def _interp_plane(planes, dim):
    """In-plane Q1 interpolation of one (or the sum of two) coarse planes, reference summation order."""
    def inplane(P):
        if dim == 2:            # a "plane" is a grid line
            out = np.zeros(2 * P.shape[0] - 1)
            out[::2] = P
            out[1::2] = P[:-1] + P[1:]
            return out, np.array([1.0 if i % 2 == 0 else 0.5 for i in range(out.size)])
        out = np.zeros((2 * P.shape[0] - 1, 2 * P.shape[1] - 1))
        out[::2, ::2] = P
        out[::2, 1::2] = P[:, :-1] + P[:, 1:]
        out[1::2, ::2] = P[:-1, :] + P[1:, :]
        out[1::2, 1::2] = P[:-1, :-1] + P[:-1, 1:] + P[1:, :-1] + P[1:, 1:]
        wgt = np.ones_like(out)
        wgt[::2, 1::2] = 0.5
        wgt[1::2, ::2] = 0.5
        wgt[1::2, 1::2] = 0.25
        return out, wgt
    if len(planes) == 1:
        s, w = inplane(planes[0])
        return np.where(w == 1.0, s, w * s)
    # odd plane: every in-plane partial sum of the lower plane comes before the upper plane's
    lo, w = _raw_terms(planes[0], dim)
    hi, _ = _raw_terms(planes[1], dim)
    s = lo[0]
    for t in lo[1:]:
        s = s + t
    for t in hi:
        s = s + t
    return (0.5 * w) * s


def _raw_terms(P, dim):
    """Corner contributions of one coarse plane to the fine in-plane nodes, x-neighbour first."""
    if dim == 2:
        n = 2 * P.shape[0] - 1
        a = np.zeros(n); b = np.zeros(n)
        a[::2] = P; a[1::2] = P[:-1]; b[1::2] = P[1:]
        w = np.ones(n); w[1::2] = 0.5
        return [a, b], w
    n0, n1 = 2 * P.shape[0] - 1, 2 * P.shape[1] - 1
    t = [np.zeros((n0, n1)) for _ in range(4)]
    t[0][::2, ::2] = P
    t[0][::2, 1::2] = P[:, :-1]; t[1][::2, 1::2] = P[:, 1:]
    t[0][1::2, ::2] = P[:-1, :]; t[1][1::2, ::2] = P[1:, :]
    t[0][1::2, 1::2] = P[:-1, :-1]; t[1][1::2, 1::2] = P[:-1, 1:]
    t[2][1::2, 1::2] = P[1:, :-1]; t[3][1::2, 1::2] = P[1:, 1:]
    w = np.ones((n0, n1)); w[::2, 1::2] = 0.5; w[1::2, ::2] = 0.5; w[1::2, 1::2] = 0.25
    return t, w
