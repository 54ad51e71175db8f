"""Multi-GPU plumbing of the pair-posterior phase: pair-index sharding and the one all-gather.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU
tests).  The N(N-1)/2 pair jobs are independent: they are sorted by cost and dealt round-robin to the
ranks, every rank runs its shard with no communication, and the sparse posteriors are
exchanged with a single all-gather of fixed-stride slabs.  After it every rank can address every
pair's rows (GatheredPairs), which is what guide-tree construction and the consistency transforms
need.  torch is used for device memory and the collective only.
"""
import numpy as np


def shard_pairs(lens, world, rank):
    """Row-major (i<j) pairs sorted by len_i*len_j descending, dealt round-robin.
    Returns (pair_x, pair_y, total_pairs) of this rank, longest first."""
    lens = np.asarray(lens, dtype=np.int64)
    n = len(lens)
    ii, jj = np.triu_indices(n, k=1)
    order = np.argsort(-(lens[ii] * lens[jj]), kind="stable")
    mine = order[rank::world]
    return ii[mine].astype(np.int64), jj[mine].astype(np.int64), len(order)


def pair_id(x, y, n):
    """row-major index of pair x<y among n sequences"""
    return x * n - x * (x + 1) // 2 + (y - x - 1)


class GatheredPairs:
    """Every rank's shard after the all-gather; csr(x, y) returns the rows of mp[x][y] for any x != y."""

    def __init__(self, lens, world, meta, rowptr, col, val, strides):
        self.lens = np.asarray(lens, dtype=np.int64)
        self.n = len(self.lens)
        self.world = world
        self.meta, self.rowptr, self.col, self.val = meta, rowptr, col, val
        self.s_pairs, self.s_rp, self.s_pool = strides
        self._where = {}
        self._rp_off = []
        for r in range(world):
            px, py, _ = shard_pairs(self.lens, world, r)
            sizes = self.lens[px] + 1 + self.lens[py] + 1
            self._rp_off.append(np.concatenate([[0], np.cumsum(sizes)]))
            for k, (x, y) in enumerate(zip(px, py)):
                self._where[(int(x), int(y))] = (r, k)

    def _entry(self, x, y):
        r, k = self._where[(x, y)]
        m = self.meta[(r * self.s_pairs + k) * 4:(r * self.s_pairs + k) * 4 + 4]
        nnz = int(m[0])
        sim = np.array([int(m[1])], np.int32).view(np.float32)[0]
        off = int(np.array([int(m[2]), int(m[3])], np.int32).view(np.int64)[0])
        return r, k, nnz, sim, off

    def sim(self, x, y):
        if x == y:
            return np.float32(1.0)
        return self._entry(min(x, y), max(x, y))[3]

    def sim_matrix(self):
        out = np.eye(self.n, dtype=np.float32)
        for (x, y) in self._where:
            out[x, y] = out[y, x] = self.sim(x, y)
        return out

    def csr(self, x, y):
        a, b = min(x, y), max(x, y)
        r, k, nnz, _, off = self._entry(a, b)
        l1, l2 = int(self.lens[a]) + 1, int(self.lens[b]) + 1
        rp0 = r * self.s_rp + int(self._rp_off[r][k])
        e0 = r * self.s_pool + off
        if x < y:
            return self.rowptr[rp0:rp0 + l1], self.col[e0:e0 + nnz], self.val[e0:e0 + nnz]
        return self.rowptr[rp0 + l1:rp0 + l1 + l2], self.col[e0 + nnz:e0 + 2 * nnz], self.val[e0 + nnz:e0 + 2 * nnz]


    def forward_arrays(self):
        """(nnz, rowptr, col, val) of all pairs in row-major order -- the arguments of Context.set_mp, i.e. how a rank
        takes the gathered shards back into its context for the consistency transforms and the progressive phase."""
        nnz, rps, cols, vals = [], [], [], []
        for x in range(self.n):
            for y in range(x + 1, self.n):
                rp, col, val = self.csr(x, y)
                nnz.append(len(col)); rps.append(rp); cols.append(col); vals.append(val)
        return (np.array(nnz, np.uint32), np.concatenate(rps).astype(np.uint32), np.concatenate(cols).astype(np.uint32),
                np.concatenate(vals).astype(np.float32))


class ShardExchange:
    """Pre-sized slabs for the all-gather of one shard's outputs (sizes exchanged once up front)."""

    def __init__(self, dist, device, world, n_pairs_local, rp_total_local, pool_cap_local):
        import torch
        self.dist, self.torch, self.world, self.device = dist, torch, world, device
        sizes = torch.tensor([n_pairs_local, rp_total_local, pool_cap_local], dtype=torch.int64, device=device)
        dist.all_reduce(sizes, op=dist.ReduceOp.MAX)
        self.s_pairs, self.s_rp, self.s_pool_cap = [int(v) for v in sizes.tolist()]
        self.n_pairs, self.rp_total = n_pairs_local, rp_total_local
        self.send_meta = torch.zeros(self.s_pairs * 4, dtype=torch.int32, device=device)
        self.send_rp = torch.zeros(self.s_rp, dtype=torch.int32, device=device)
        self.recv_meta = torch.empty(world * self.s_pairs * 4, dtype=torch.int32, device=device)
        self.recv_rp = torch.empty(world * self.s_rp, dtype=torch.int32, device=device)
        self.recv_col = self.recv_val = None
        self.s_pool = 0

    def exchange(self, pair_nnz, sim, pair_off, rowptr, col, val, pool_used):
        """pair_nnz int32[n], sim float32[n], pair_off int64[n], rowptr int32[rp_total], col int32[cap],
        val float32[cap] (all on self.device); pool_used = entries this rank produced (None: as in the previous
        call).  One max-reduce for the payload stride, then the gather itself."""
        torch, dist = self.torch, self.dist
        if pool_used is not None or self.s_pool == 0:
            mx = torch.tensor([int(pool_used)], dtype=torch.int64, device=self.device)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            self.s_pool = int(mx.item())
        # pool_used=None: a repeat of an exchange whose sizes are known (same inputs): the stride agreed then is
        # reused and nothing in this call waits for the device
        m = self.send_meta.view(self.s_pairs, 4)
        m[:self.n_pairs, 0] = pair_nnz
        m[:self.n_pairs, 1] = sim.view(torch.int32)
        m[:self.n_pairs, 2:4] = pair_off.view(torch.int32).view(self.n_pairs, 2)
        self.send_rp[:self.rp_total] = rowptr[:self.rp_total]
        if self.recv_col is None or self.recv_col.numel() != self.world * self.s_pool:
            self.recv_col = torch.empty(self.world * self.s_pool, dtype=torch.int32, device=self.device)
            self.recv_val = torch.empty(self.world * self.s_pool, dtype=torch.float32, device=self.device)
        send_col, send_val = col[:self.s_pool], val[:self.s_pool]
        if send_col.numel() < self.s_pool:  # this rank's pool is shorter than the longest: pad
            send_col = torch.cat([send_col, torch.zeros(self.s_pool - send_col.numel(), dtype=torch.int32, device=self.device)])
            send_val = torch.cat([send_val, torch.zeros(self.s_pool - send_val.numel(), dtype=torch.float32, device=self.device)])
        dist.all_gather_into_tensor(self.recv_meta, self.send_meta)
        dist.all_gather_into_tensor(self.recv_rp, self.send_rp)
        dist.all_gather_into_tensor(self.recv_col, send_col.contiguous())
        dist.all_gather_into_tensor(self.recv_val, send_val.contiguous())

    def gathered(self, lens):
        return GatheredPairs(lens, self.world, self.recv_meta.cpu().numpy(), self.recv_rp.cpu().numpy().view(np.uint32),
                             self.recv_col.cpu().numpy().view(np.uint32), self.recv_val.cpu().numpy(),
                             (self.s_pairs, self.s_rp, self.s_pool))
