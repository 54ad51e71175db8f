"""Multi-GPU plumbing: pair-index sharding and the gathers of the sparse posteriors.

Two users.  bench.py times phase 1 at the L0 (device pointer) level: cost-sorted deal of the pair jobs
(shard_pairs) and ONE all-gather of a packed slab per step (ShardExchange).  phase1_sharded is the whole
sharded phase 1 of a run on top of the L1 C ABI: folds by x mod G, pair posteriors and the matching
consistency transform by contiguous pair-index ranges, each followed by a gather; after it every rank's
context holds the complete stores and the rest of the run (guide tree, progressive phase) is replicated
(phase 2 is tree-sequential).

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
    """The one exchange of phase 1 at the L0 level: every rank's {per-pair nnz / sim / pool offset, row pointers,
    entry columns, entry values} travel as ONE packed int32 slab in ONE all-gather (fixed strides agreed once up
    front; a rank whose parts are shorter pads).  Slab of rank r: [4 * s_pairs meta | s_rp row pointers | s_pool
    columns | s_pool values (bits)]."""

    def __init__(self, dist, device, world, n_pairs_local, rp_total_local, pool_cap_local):
        import torch
        self.dist, self.torch, self.world, self.device = dist, torch, world, device
        sizes = torch.tensor([n_pairs_local, rp_total_local, pool_cap_local], dtype=torch.int64, device=device)
        dist.all_reduce(sizes, op=dist.ReduceOp.MAX)
        self.s_pairs, self.s_rp, self.s_pool_cap = [int(v) for v in sizes.tolist()]
        self.n_pairs, self.rp_total = n_pairs_local, rp_total_local
        self.s_pool = 0
        self.send = self.recv = None

    def _layout(self):
        m, r, c = 4 * self.s_pairs, self.s_rp, self.s_pool
        return m, r, c, m + r + 2 * c

    def exchange(self, pair_nnz, sim, pair_off, rowptr, col, val, pool_used):
        """pair_nnz int32[n], sim float32[n], pair_off int64[n], rowptr int32[rp_total], col int32[cap],
        val float32[cap] (all on self.device); pool_used = entries this rank produced (None: as in the previous
        call).  One max-reduce for the payload stride the first time, then the gather itself."""
        torch, dist = self.torch, self.dist
        if pool_used is None and self.s_pool == 0:
            raise ValueError("ShardExchange.exchange: the first call needs pool_used (the payload stride is agreed from it)")
        if pool_used is not None:
            mx = torch.tensor([int(pool_used)], dtype=torch.int64, device=self.device)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            self.s_pool = int(mx.item())
        # pool_used=None: a repeat of an exchange whose sizes are known (same inputs): the stride agreed then is
        # reused and nothing in this call waits for the device
        m, r, c, total = self._layout()
        if self.send is None or self.send.numel() != total:
            self.send = torch.zeros(total, dtype=torch.int32, device=self.device)
            self.recv = torch.empty(self.world * total, dtype=torch.int32, device=self.device)
        meta = self.send[:m].view(self.s_pairs, 4)
        meta[:self.n_pairs, 0] = pair_nnz
        meta[:self.n_pairs, 1] = sim.view(torch.int32)
        meta[:self.n_pairs, 2:4] = pair_off.view(torch.int32).view(self.n_pairs, 2)
        self.send[m:m + self.rp_total] = rowptr[:self.rp_total]
        k = min(c, col.numel())
        self.send[m + r:m + r + k] = col[:k]
        self.send[m + r + c:m + r + c + k] = val[:k].view(torch.int32)
        dist.all_gather_into_tensor(self.recv, self.send)

    def gathered(self, lens):
        m, r, c, total = self._layout()
        rows = self.recv.cpu().numpy().reshape(self.world, total)
        meta = np.ascontiguousarray(rows[:, :m]).reshape(-1)
        rp = np.ascontiguousarray(rows[:, m:m + r]).reshape(-1).view(np.uint32)
        col = np.ascontiguousarray(rows[:, m + r:m + r + c]).reshape(-1).view(np.uint32)
        val = np.ascontiguousarray(rows[:, m + r + c:]).reshape(-1).view(np.float32)
        return GatheredPairs(lens, self.world, meta, rp, col, val, (self.s_pairs, self.s_rp, self.s_pool))


# ---------------------------------------------------------------------------------------------
# sharded phase 1 of a whole run (L1 C ABI + torch.distributed; gloo in the tests, nccl = RCCL on a node)
# ---------------------------------------------------------------------------------------------
def _comm_device(dist, device):
    import torch
    return device if dist.get_backend() == "nccl" else torch.device("cpu")


def allgather_concat(dist, arr, device):
    """concatenation over ranks (in rank order) of 1-D numpy arrays of different lengths: one size exchange, one
    all-gather of max-padded rows"""
    import torch
    arr = np.ascontiguousarray(arr)
    world = dist.get_world_size()
    dev = _comm_device(dist, device)
    n = torch.tensor([arr.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(v.item()) for v in sizes]
    mx = max(sizes + [1])
    raw = np.zeros(mx * arr.itemsize, np.uint8)
    raw[:arr.nbytes] = arr.view(np.uint8).reshape(-1)
    send = torch.from_numpy(raw).to(dev)
    recv = torch.empty(world * raw.size, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send)
    rows = recv.cpu().numpy().reshape(world, raw.size)
    return np.concatenate([rows[r, :sizes[r] * arr.itemsize].view(arr.dtype) for r in range(world)])


def allreduce_sum(dist, arr, device):
    """elementwise sum over ranks of equally shaped integer arrays (shards that are zero outside their own part)"""
    import torch
    dev = _comm_device(dist, device)
    t = torch.from_numpy(np.ascontiguousarray(arr).astype(np.int64)).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(arr.dtype)


def pair_ranges(npairs, world):
    """contiguous row-major pair-index ranges, one per rank (what dafs_hip_align_posteriors takes)"""
    return [npairs * k // world for k in range(world + 1)]


def gather_parts(dist, parts, device):
    """The one collective of an exchange: every rank contributes a list of 1-D 4-byte tensors on `device` (the same number
    on every rank, lengths differ); returns, per part, the concatenation over the ranks in rank order -- on the device.
    One tiny all-gather of the lengths (the strides have to be agreed), then ONE all_gather_into_tensor of the packed,
    max-padded slab.  With the nccl backend (RCCL over xGMI) nothing touches host memory; gloo, which the CPU-side tests
    use, carries the slab through the host."""
    import torch
    world = dist.get_world_size()
    nccl = dist.get_backend() == "nccl"
    cdev = device if nccl else torch.device("cpu")
    mine = torch.tensor([int(p.numel()) for p in parts], dtype=torch.int64, device=cdev)
    allsz = torch.empty(world * len(parts), dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(allsz, mine)
    sizes = allsz.cpu().numpy().reshape(world, len(parts))
    strides = sizes.max(axis=0)
    offs = np.concatenate([[0], np.cumsum(strides)])
    total = int(offs[-1])
    if total == 0:
        return [p[:0] for p in parts]
    slab = torch.zeros(total, dtype=torch.int32, device=device)
    for k, p in enumerate(parts):
        if p.numel():
            slab[int(offs[k]):int(offs[k]) + p.numel()] = p.view(torch.int32)
    recv = torch.empty(world * total, dtype=torch.int32, device=cdev)
    dist.all_gather_into_tensor(recv, slab if nccl else slab.cpu())
    if not nccl:
        recv = recv.to(device)
    out = []
    for k, p in enumerate(parts):
        segs = [recv[r * total + int(offs[k]):r * total + int(offs[k]) + int(sizes[r, k])] for r in range(world)]
        out.append(torch.cat(segs).view(p.dtype))
    # the results go to the library next, which works on a stream of its own: what torch has queued (copies, concatenations)
    # must have landed before it reads them
    if device.type == "cuda":
        torch.cuda.current_stream(device).synchronize()
    return out


def phase1_sharded(ctx, seqs, dist, device, align_model, th_a, w_pct_a, w_pct_s, fold_th=0.01):
    """Phase 1 of DAFS::run (dafs.cpp:1787-1827) on `world` ranks, one context (one GPU) each, device-resident:
      * base-pairing posteriors of the sequences x = rank (mod world): exported to device buffers, ONE all-gather, installed
        by block (dafs_hip_bp_export_dev / dafs_hip_set_bp_dev);
      * pair posteriors + similarity scores of this rank's contiguous pair-index range: ONE all-gather, installed
        (dafs_hip_mp_export_dev / _install_dev) -- concatenating the ranks' ranges in rank order IS the whole in pair order;
      * relax_matching_probability for this rank's range of OUTPUT pairs: ONE more all-gather, installed as the relaxed store;
      * relax_basepairing_probability replicated (milliseconds).
    Afterwards ctx is in the state a single-GPU phase 1 leaves it in, bit for bit.  torch supplies the buffers and the
    collective (gather_parts); no array of the stores passes through host memory on the nccl backend."""
    import torch
    from . import capi
    rank, world = dist.get_rank(), dist.get_world_size()
    n = len(seqs)
    npairs = n * (n - 1) // 2
    i32 = lambda k: torch.empty(max(int(k), 1), dtype=torch.int32, device=device)
    f32 = lambda k: torch.empty(max(int(k), 1), dtype=torch.float32, device=device)
    ctx.set_sequences(seqs)
    # ---- folds: x = rank mod world, in a context of their own ----
    mine = list(range(rank, n, world))
    rp_t, col_t, val_t = i32(0)[:0], i32(0)[:0], f32(0)[:0]
    if mine:
        fc = capi.Context(ctx.device_index)
        try:
            fc.set_sequences([seqs[x] for x in mine])
            fc.fold_posteriors(fold_th)
            ne, nr = fc.bp_sizes(0)
            rp_b, col_b, val_b = i32(nr), i32(ne), f32(ne)
            nr, ne = fc.bp_export_dev(rp_b.data_ptr(), col_b.data_ptr(), val_b.data_ptr(), ne)
            rp_t, col_t, val_t = rp_b[:nr], col_b[:ne], val_b[:ne]
        finally:
            fc.close()
    rp_g, col_g, val_g = gather_parts(dist, [rp_t, col_t, val_t], device)
    order = [x for r in range(world) for x in range(r, n, world)]  # sequence of the k-th gathered block
    ctx.set_bp_dev(order, rp_g.data_ptr(), col_g.data_ptr(), val_g.data_ptr(), int(col_g.numel()))
    # ---- pair posteriors of [b0, b1) ----
    b = pair_ranges(npairs, world)
    cnt = b[rank + 1] - b[rank]
    parts = [i32(0)[:0], i32(0)[:0], i32(0)[:0], f32(0)[:0], f32(0)[:0]]
    if cnt:
        ctx.align_posteriors(align_model, th_a, pair_begin=b[rank], pair_end=b[rank + 1], fetch=False)
        _, ne, nr = ctx.mp_sizes(0)
        nnz_b, rp_b, col_b, val_b, sim_b = i32(cnt), i32(nr), i32(ne), f32(ne), f32(cnt)
        nr, ne = ctx.mp_export_dev(0, 0, cnt, nnz_b.data_ptr(), rp_b.data_ptr(), col_b.data_ptr(), val_b.data_ptr(), sim_b.data_ptr(), ne)
        parts = [nnz_b[:cnt], rp_b[:nr], col_b[:ne], val_b[:ne], sim_b[:cnt]]
    nnz_g, rp_g, col_g, val_g, sim_g = gather_parts(dist, parts, device)
    ctx.mp_install_dev(0, nnz_g.data_ptr(), rp_g.data_ptr(), col_g.data_ptr(), val_g.data_ptr(), sim_g.data_ptr(), int(col_g.numel()))
    # ---- consistency: matching transform sharded by output pair, base-pairing transform replicated ----
    if w_pct_a != 0.0:
        parts = [i32(0)[:0], i32(0)[:0], i32(0)[:0], f32(0)[:0]]
        if cnt:
            ctx.consistency_match_range(w_pct_a, b[rank], b[rank + 1])
            _, ne, nr = ctx.mp_sizes(1)          # entries of the shard; row pointers of all pairs (the range's are exported)
            nnz_b, rp_b, col_b, val_b = i32(cnt), i32(nr), i32(ne), f32(ne)
            nr, ne = ctx.mp_export_dev(1, b[rank], cnt, nnz_b.data_ptr(), rp_b.data_ptr(), col_b.data_ptr(), val_b.data_ptr(), None, ne)
            parts = [nnz_b[:cnt], rp_b[:nr], col_b[:ne], val_b[:ne]]
        nnz_g, rp_g, col_g, val_g = gather_parts(dist, parts, device)
        ctx.mp_install_dev(1, nnz_g.data_ptr(), rp_g.data_ptr(), col_g.data_ptr(), val_g.data_ptr(), None, int(col_g.numel()))
    if w_pct_s != 0.0:
        ctx.consistency_bp(w_pct_s)


def phase1_sharded_host(ctx, seqs, dist, device, align_model, th_a, w_pct_a, w_pct_s, fold_th=0.01):
    """The same exchange through host arrays (round 2's form, kept as the cross-check of the device-resident one in
    tests/test_dist_gpu.py)."""
    from . import capi
    rank, world = dist.get_rank(), dist.get_world_size()
    n = len(seqs)
    npairs = n * (n - 1) // 2
    ctx.set_sequences(seqs)
    # ---- folds: x = rank mod world, in a context of their own; rows gathered as three flat arrays ----
    mine = list(range(rank, n, world))
    rows = []
    if mine:
        fc = capi.Context(ctx.device_index)
        try:
            fc.set_sequences([seqs[x] for x in mine])
            fc.fold_posteriors(fold_th)
            rows = fc.bp(0)
        finally:
            fc.close()
    rp = allgather_concat(dist, np.concatenate([r[0] for r in rows]) if rows else np.zeros(0, np.uint32), device)
    col = allgather_concat(dist, np.concatenate([r[1] for r in rows]) if rows else np.zeros(0, np.uint32), device)
    val = allgather_concat(dist, np.concatenate([r[2] for r in rows]) if rows else np.zeros(0, np.float32), device)
    order = [x for r in range(world) for x in range(r, n, world)]  # sequence of the k-th gathered row block
    all_rows = [None] * n
    r0 = e0 = 0
    for x in order:
        L = len(seqs[x])
        r = rp[r0:r0 + L + 1]
        k = int(r[-1])
        all_rows[x] = (r, col[e0:e0 + k], val[e0:e0 + k])
        r0 += L + 1
        e0 += k
    ctx.set_bp(all_rows)
    # ---- pair posteriors of [b0, b1): concatenating the ranks' arrays in rank order IS the whole in pair order ----
    b = pair_ranges(npairs, world)
    res = ctx.align_posteriors(align_model, th_a, pair_begin=b[rank], pair_end=b[rank + 1]) if b[rank + 1] > b[rank] else None
    z32, zf = np.zeros(0, np.uint32), np.zeros(0, np.float32)
    nnz = allgather_concat(dist, res.nnz if res else z32, device)
    rowptr = allgather_concat(dist, res._rowptr if res else z32, device)
    col = allgather_concat(dist, res._col if res else z32, device)
    val = allgather_concat(dist, res._val if res else zf, device)
    sim = allgather_concat(dist, res.sim if res else zf, device)
    ctx.mp_install(0, nnz, rowptr, col, val, sim)
    # ---- consistency: matching transform sharded by output pair, base-pairing transform replicated ----
    if w_pct_a != 0.0:
        if b[rank + 1] > b[rank]:
            ctx.consistency_match_range(w_pct_a, b[rank], b[rank + 1])
            part = ctx.mp(1)
            p_nnz, p_rp, p_col, p_val = part.nnz, part._rowptr, part._col, part._val
        else:
            lens = np.array([len(s) for s in seqs], np.int64)
            ii, jj = np.triu_indices(n, k=1)
            p_nnz, p_rp, p_col, p_val = np.zeros(npairs, np.uint32), np.zeros(int((lens[ii] + lens[jj] + 2).sum()), np.uint32), z32, zf
        nnz = allreduce_sum(dist, p_nnz, device)      # zero outside the shard
        rowptr = allreduce_sum(dist, p_rp, device)
        col = allgather_concat(dist, p_col, device)   # shard entries only, shards are in pair order
        val = allgather_concat(dist, p_val, device)
        ctx.mp_install(1, nnz, rowptr, col, val)
    if w_pct_s != 0.0:
        ctx.consistency_bp(w_pct_s)
