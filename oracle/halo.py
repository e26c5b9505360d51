"""TEST INFRASTRUCTURE ONLY -- numpy restatement of atlas::parallel::HaloExchange for a set of ranks emulated in one
process.  Reference: src/atlas/parallel/HaloExchange.cc:78-172 (setup), HaloExchange.h:151-290 (execute /
execute_adjoint), detail/pack_index.h + Packer.cc:19-37 (buffer order: for node in map: non-parallel indices
row-major), detail/adjoint_unpack_index.h / zero_index.h (adjoint: field += buffer, then halos zeroed)."""
import numpy as np


class HaloExchangeOracle:
    """one object per rank; `ranks` is the list of all of them (index == MPI rank)"""

    def __init__(self, rank, nproc):
        self.rank, self.nproc = rank, nproc

    def setup_local(self, part, ridx, base, parsize, halo_begin=0):
        part, ridx = np.asarray(part), np.asarray(ridx)
        self.parsize = parsize
        idx = np.arange(halo_begin, parsize)
        ghost = idx[(part[idx] != self.rank) | (ridx[idx] != base + idx)]   # IsGhostPoint, HaloExchange.cc:31-52
        self.recvcounts = np.bincount(part[ghost], minlength=self.nproc).astype(np.int64)
        self.recvdispls = np.concatenate([[0], np.cumsum(self.recvcounts)[:-1]])
        order = np.argsort(part[ghost], kind="stable")                      # grouped by owner, ascending index
        self.recvmap = ghost[order]
        self.send_requests = (ridx[self.recvmap] - base).astype(np.int64)
        self.recvcnt = int(self.recvcounts.sum())

    @staticmethod
    def setup(ranks, parts, ridxs, base, parsizes, halo_begins=None):
        n = len(ranks)
        for r in range(n):
            ranks[r].setup_local(parts[r], ridxs[r], base, parsizes[r], 0 if halo_begins is None else halo_begins[r])
        for r in range(n):  # allToAll of counts + allToAllv of requests (HaloExchange.cc:118,156-159)
            me = ranks[r]
            me.sendcounts = np.array([ranks[p].recvcounts[r] for p in range(n)], dtype=np.int64)
            me.senddispls = np.concatenate([[0], np.cumsum(me.sendcounts)[:-1]])
            me.sendmap = np.concatenate([
                ranks[p].send_requests[ranks[p].recvdispls[r]:ranks[p].recvdispls[r] + ranks[p].recvcounts[r]]
                for p in range(n)]).astype(np.int64) if me.sendcounts.sum() else np.zeros(0, dtype=np.int64)
            me.sendcnt = int(me.sendcounts.sum())

    @staticmethod
    def _nodes_first(field, parallel_dim):
        return np.moveaxis(field, parallel_dim, 0)

    @staticmethod
    def execute(ranks, fields, parallel_dim=0):
        """fields: list of numpy arrays (views allowed), one per rank; modified in place"""
        n = len(ranks)
        send = [HaloExchangeOracle._nodes_first(fields[r], parallel_dim)[ranks[r].sendmap].copy() for r in range(n)]
        for r in range(n):
            me = ranks[r]
            v = HaloExchangeOracle._nodes_first(fields[r], parallel_dim)
            for p in range(n):
                c = me.recvcounts[p]
                if c:
                    src = send[p][ranks[p].senddispls[r]:ranks[p].senddispls[r] + c]
                    v[me.recvmap[me.recvdispls[p]:me.recvdispls[p] + c]] = src

    @staticmethod
    def execute_adjoint(ranks, fields, parallel_dim=0):
        n = len(ranks)
        halo = [HaloExchangeOracle._nodes_first(fields[r], parallel_dim)[ranks[r].recvmap].copy() for r in range(n)]
        for r in range(n):
            me = ranks[r]
            v = HaloExchangeOracle._nodes_first(fields[r], parallel_dim)
            for p in range(n):
                c = me.sendcounts[p]
                if c:
                    src = halo[p][ranks[p].recvdispls[r]:ranks[p].recvdispls[r] + c]
                    np.add.at(v, me.sendmap[me.senddispls[p]:me.senddispls[p] + c], src)
        for r in range(n):
            HaloExchangeOracle._nodes_first(fields[r], parallel_dim)[ranks[r].recvmap] = 0
