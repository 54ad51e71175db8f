#!/usr/bin/env python3
"""bench.py -- throughput of the DAFS pair-posterior hot path on MI355X.

Metric (BASELINE.json): seq-pairs/sec on N=128, L~150 synthetic random RNA.
One "step" = one pass of the all-pairs ProbCons pair-HMM path over the whole batch, inputs
resident in HBM: forward/backward/posterior, thresholding, sparse rows of mp[x][y] and mp[y][x],
and the similarity score of every pair (reference src/align.cpp:35-79,
src/probconsRNA/ProbabilisticModel.h:105-403, src/dafs.cpp:155-167,713-764).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

With N > 1 the pairs of the SAME set (BASELINE.json's metric is N=128, L~150 at 1/2/4/8 GPUs: strong scaling) are dealt
to the ranks by cost, each rank runs its shard, and the sparse posteriors are exchanged with one all-gather over RCCL
(the only collective of the path).  --scaling weak grows the set by sqrt(N) instead (~8128 pairs per GPU);
--config c2|c3|c4|c5 picks another BASELINE configuration (c4 = N=256, L~200, the multi-GPU config).

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
"roofline" (algorithmic bytes / kernel time against the 8 TB/s HBM peak) and "cpu_baseline"
(the CPU path timed on this box's host cores on a bounded sample of the same pairs).
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# Vector issue ceiling for arithmetic that must stay bit-exact with a non-contracting CPU build: plain (unpacked, unfused)
# FP32 / integer lane-operations, 256 CUs x 4 SIMDs x 16 lanes per clock x 2.4 GHz.  The 157.3 TFLOP/s spec figure counts a
# packed FMA as four flops per lane slot; neither packing nor FMA contraction applies to an ordered float sum.
SIMDS, CLOCK_HZ = 1024, 2.4e9
VALU_LANE_OPS = SIMDS * 16 * CLOCK_HZ
BASE_N, BASE_L = 128, 150
CONFIGS = {"c2": (32, 80), "c3": (128, 150), "c4": (256, 200), "c5": (512, 400)}  # BASELINE.json configs (N, L)


def build_shard(n_seq, length, world, rank, seed=12345):
    from dafs_amd import dist as ddist, synth
    recs = synth.random_set(n_seq, length, seed=seed)
    names = [n for n, _ in recs]
    seqs = [s for _, s in recs]
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    px, py, total = ddist.shard_pairs(lens, world, rank)
    return names, seqs, lens, px, py, total


def _pair_order(npairs, step=61):
    # stride through the cost-sorted list so that any prefix of the sample spans all lengths
    return [k for r in range(step) for k in range(r, npairs, step)]


def _cpu_pairs_worker(job):
    """one worker of the CPU baseline: its share of the sample, timed inside the worker"""
    seqs, pairs, th, model, budget_s = job
    import oracle_lib
    orc = oracle_lib.load_oracle()
    ref = oracle_lib.load_ref() if model == 0 else None
    calc = ref.align_calculate if ref is not None else (lambda a, b, t: orc.align_calculate(a, b, t, model))
    n = 0
    t0 = time.perf_counter()
    for x, y in pairs:
        a, b = seqs[x], seqs[y]
        rp, col, val = calc(a, b, th)
        orc.similarity(rp, col, val, len(a), len(b))
        rows = np.repeat(np.arange(len(a), dtype=np.uint32), np.diff(rp))  # transpose_mp
        o = np.lexsort((rows, col))
        _ = rows[o], val[o]
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    return n, time.perf_counter() - t0, ref is not None


def cpu_baseline(seqs, px, py, th, model, length, budget_s=10.0):
    """CPU path on a bounded sample of the pairs: one core (the reference has no threading), then process-parallel
    over pairs on the host cores this job may use.
    kind "reference": the reference's own ProbCons::calculate (oracle/_ref, built from /root/reference by
    oracle/Makefile) for the posterior + sparse rows, plus the oracle's restatement of transpose_mp /
    calculate_similarity_score (dafs.cpp is not compilable here).  kind "port": the oracle restatement for everything
    (CONTRAlign runs, or boxes without oracle/_ref).  Must run before this process touches the GPU (it forks)."""
    import multiprocessing as mp
    order = _pair_order(len(px))
    pairs = [(int(px[k]), int(py[k])) for k in order]
    n1, dt1, is_ref = _cpu_pairs_worker((seqs, pairs, th, model, budget_s))
    cores = min(len(os.sched_getaffinity(0)), 16)
    per = (len(pairs) + cores - 1) // cores
    jobs = [(seqs, pairs[w::cores][:per], th, model, budget_s) for w in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(cores) as pool:
        res = pool.map(_cpu_pairs_worker, jobs)
    wall = time.perf_counter() - t0
    n_all = sum(r[0] for r in res)
    return {"value": n1 / dt1, "unit": "seq-pairs/s", "cores": 1, "kind": "reference" if is_ref else "port",
            "sample": "%d of the %d pairs (N=%d L~%d set), posterior+sparse rows+transpose+sim, %.1f s" % (n1, len(px), len(seqs), length, dt1),
            "all_cores": {"value": n_all / wall, "unit": "seq-pairs/s", "cores": cores,
                          "sample": "%d pairs over %d worker processes, %.1f s wall" % (n_all, cores, wall)}}


def cpu_end_to_end(names, seqs, model):
    """the oracle's whole run (oracle/pipeline.c, one core) on the same set: wall-clock and phase split"""
    import oracle_lib
    orc = oracle_lib.load_oracle()
    pl = orc.pipeline(names, seqs, orc.params(fold_model=0, align_model=model))
    t0 = time.perf_counter()
    pl.phase1()
    t1 = time.perf_counter()
    pl.phase2()
    t2 = time.perf_counter()
    out = pl.output()
    sec = pl.seconds()
    pl.close()
    return {"wall_s": t2 - t0, "cores": 1, "kind": "port",
            "phases_s": {"fold": round(sec[0], 3), "pair": round(sec[1], 3), "pct_tree": round(sec[2], 3), "progressive": round(sec[3], 3)},
            "phase1_s": round(t1 - t0, 3), "phase2_s": round(t2 - t1, 3)}, out


def cold_cli(names, seqs, contra):
    """the drop-in itself: dafs_amd/dafs on the FASTA of the set, one cold process (start-up, context, code objects,
    every allocation) -- what the reference's 30-45 s wall-clock compares with"""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "dafs_amd", "dafs")
    if not os.path.exists(exe):
        return None
    with tempfile.NamedTemporaryFile("w", suffix=".fa", delete=False) as f:
        for n, sq in zip(names, seqs):
            f.write(">%s\n%s\n" % (n, sq))
        path = f.name
    cmd = [exe, "-s", "CONTRAfold", "--no-alifold"] + (["-a", "CONTRAlign"] if contra else []) + [path]
    t0 = time.perf_counter()
    r = subprocess.run(cmd, capture_output=True, text=True)
    wall = time.perf_counter() - t0
    os.unlink(path)
    if r.returncode != 0:
        return {"error": r.stderr.strip()[-200:]}
    return {"wall_s": wall, "command": "dafs -s CONTRAfold --no-alifold%s FASTA" % (" -a CONTRAlign" if contra else ""), "stdout": r.stdout}


def _pmc_file(kernel_name):
    """the newest committed PMC summary (profiles/*_pmc.json, tools/pmc_summary.py) of this kernel instance and of these kernel sources"""
    import glob
    from dafs_amd import build
    sha = build.source_sha16()  # a summary counts only for the kernel sources it measured
    hit = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
        try:
            pm = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        if isinstance(pm.get("kernel"), str) and pm["kernel"].startswith(kernel_name) and pm.get("kernel_source_sha16") == sha:
            hit = (os.path.basename(f), pm)
    return hit


def _stage(ms, launches, alg_bytes=None, **more):
    d = {"ms": round(ms, 4), "launches": launches}
    if alg_bytes is not None:
        d["algorithmic_bytes"] = float(alg_bytes)
        d["achieved_GBps"] = alg_bytes / (ms * 1e-3) / 1e9 if ms else None
        d["frac"] = d["achieved_GBps"] / HBM_PEAK_GBS if ms else None
    d.update(more)
    return d


def stage_objects(ctx, seqs, lens, contra, rep, iso):
    """bench line "stages": one entry per stage kernel of the whole run, each with its device time (HIP events around the
    launches, summed; kernels on different streams overlap), SURVEY 8(d)'s algorithmic bytes, the fraction of the HBM peak,
    and the bound that actually binds.  rep: Context.stage_report() of one whole run; iso: of one isolated node."""
    n = len(seqs)
    st = {}
    L = lens.astype(np.float64)
    if "k_contrafold" in rep:  # inside 3 tables + outside 3 tables + posterior = 28*S B per sequence, + F5 arrays 8(L+1)
        ms, _, k = rep["k_contrafold"]
        alg = float((28.0 * (L + 1) * (L + 2) / 2 + 8.0 * (L + 1)).sum())
        st["k_contrafold"] = _stage(ms, k, alg, unit="28*S + 8(L+1) B per sequence, S=(L+1)(L+2)/2",
                                    binding="latency: one workgroup per sequence, a span waits for its longest chain of log-sum-exps",
                                    sequences=n, us_per_sequence_chain=ms * 1e3)
    pp = ctx.mp(0)  # the un-relaxed store: what the transform reads
    rowlen = [np.zeros((int(lens[z]), n), np.int64) for z in range(n)]  # rowlen[z][k, x] = entries of row k of mp[z][x]
    for z in range(n):
        rowlen[z][:, z] = 1  # identity (align.cpp:42-44)
    nnz_total = 0
    for p in range(len(pp)):
        x, y = int(pp.pair_x[p]), int(pp.pair_y[p])
        rowlen[x][:, y] = np.diff(pp.csr(p)[0].astype(np.int64))
        rowlen[y][:, x] = np.diff(pp.csr(p, transposed=True)[0].astype(np.int64))
        nnz_total += int(pp.nnz[p])
    # relax_matching_probability (dafs.cpp:258-324): output pair (x, y) adds, over z and k, rowlen[z][k, x] * rowlen[z][k, y] products
    addends = float(sum(((r.sum(axis=1) ** 2 - (r ** 2).sum(axis=1)).sum()) // 2 for r in rowlen))
    lsum, lsq = float(L.sum()), float((L * L).sum())
    pct_bytes = 8.0 * (n - 1) * (2.0 * nnz_total + lsum) + 4.0 * (lsum * lsum - lsq) / 2  # reads 2N CSR matrices per output pair + its dense tile
    if "k_pct_rows" in rep:
        ms, _, k = rep["k_pct_rows"]
        lane_ops = 3.0 * addends  # multiply by w, multiply the two probabilities, add: none fusable (ordered float sum, no contraction)
        st["k_pct_rows"] = _stage(ms, k, pct_bytes, unit="sum over output pairs of 8 B x entries of the 2N matrices read + 4*L1*L2 B written",
                                  addends=addends, lane_ops=lane_ops, lane_ops_per_s=lane_ops / (ms * 1e-3) if ms else None,
                                  valu_peak_lane_ops_per_s=VALU_LANE_OPS, frac_of_valu_peak=lane_ops / (ms * 1e-3) / VALU_LANE_OPS if ms else None,
                                  binding="vector issue + gather latency (the addends are 3 plain lane-operations each; the stores stay cache-resident)")
    for kname, per_cell in (("k_pairhmm3", 28.0), ("k_pairhmm5", 44.0)):
        if kname in rep:
            ms, _, k = rep[kname]
            alg = per_cell * float((lsum + n) ** 2 - ((L + 1) ** 2).sum()) / 2  # sum over pairs of (L1+1)(L2+1)
            st[kname + " (inside the whole run)"] = _stage(ms, k, alg, unit="%g*(L1+1)(L2+1) B per pair" % per_cell, binding="vector issue (see roofline.secondary)")
    if "k_dd_solve" in rep:
        ms, longest, k = rep["k_dd_solve"]
        d = _stage(ms, k, None, longest_launch_ms=round(longest, 4),
                   note="summed over the launches of both lanes of every round (they overlap); the guide tree serialises the nodes",
                   binding="latency: one wavefront per subproblem, ~3 dependent DPs per iteration")
        if iso and "k_dd_solve" in iso[0]:
            ims, _, _ = iso[0]["k_dd_solve"]
            its, l1, l2, ncbp = iso[1]
            alg_it = 16.0 * (l1 * (l1 - 1) // 2 + l2 * (l2 - 1) // 2) + 13.0 * (l1 + 1) * (l2 + 1) + 64.0 * ncbp
            d.update(isolated_node={"columns": [l1, l2], "iterations": its, "ms": round(ims, 4), "us_per_node_iteration": ims * 1e3 / its,
                                    "algorithmic_bytes_per_iteration": alg_it, "achieved_GBps": alg_it * its / (ims * 1e-3) / 1e9,
                                    "frac": alg_it * its / (ims * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "note": "one node of two single sequences alone on the device, forced to t_max iterations in one launch"})
        st["k_dd_solve"] = d
    for kname in ("k_pct_bp_rows", "k_pct_emit", "k_contrafold_posterior", "k_node_avg", "k_node_lists", "k_node_cbp_fill", "k_nussinov_single"):
        if kname in rep:
            st[kname] = _stage(rep[kname][0], rep[kname][2])
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c3", help="BASELINE.json configuration (c3 = the metric's N=128, L~150)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="--gpus N > 1: strong = the same set on N GPUs (the metric), weak = a sqrt(N)-times larger set")
    ap.add_argument("--n-seq", type=int, default=0, help="override the number of sequences")
    ap.add_argument("--length", type=int, default=0, help="override the nominal length")
    ap.add_argument("--th", type=float, default=0.01)
    ap.add_argument("--model", choices=("probcons", "contralign"), default="probcons", help="alignment model of the timed kernel")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-verify", action="store_true", help="tuning builds only (tools/build_exp.sh: kernels that stop after a sweep): skip the check of the "
                    "timed launch against the oracle; the JSON line then says verified_pairs 0")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end wall-clock legs (warm driver run, cold CLI, forced-iteration DD run)")
    args = ap.parse_args()
    cfg_n, cfg_l = CONFIGS[args.config]
    args.length = args.length or cfg_l

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): launch with torch.distributed.run" % (world, args.gpus))
    contra = args.model == "contralign"
    n_seq = args.n_seq or (int(round(cfg_n * math.sqrt(world))) if args.scaling == "weak" else cfg_n)
    names, seqs, lens, px, py, total_pairs = build_shard(n_seq, args.length, world, rank)
    np_local = len(px)

    # ---- CPU legs first: they fork worker processes, which must happen before this process touches the GPU ----
    cpu = cpu_e2e = cli = None
    cpu_out = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(seqs, px, py, args.th, 1 if contra else 0, args.length)
        if not args.no_e2e:
            cpu_e2e, cpu_out = cpu_end_to_end(names, seqs, 1 if contra else 0)
            cpu["end_to_end"] = cpu_e2e
    if rank == 0 and world == 1 and not args.no_e2e:
        cli = cold_cli(names, seqs, contra)  # a child process with a GPU context of its own

    import torch
    import torch.distributed as dist
    from dafs_amd import capi

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("DAFS_BENCH_FORCE_EXCHANGE") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import datetime
        # a collective that a rank never joins ends the run after three minutes instead of the backend's default half hour
        limit = datetime.timedelta(seconds=180)
        if world == 1:
            dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1, timeout=limit)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=limit)

    # ---- device-resident inputs (torch = allocator + stream only) ----
    codes = np.concatenate([capi.encode(s) for s in seqs])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    tasks = np.zeros((np_local, 4), np.uint32)
    tasks[:, 0] = off[px]; tasks[:, 1] = lens[px]; tasks[:, 2] = off[py]; tasks[:, 3] = lens[py]
    rp_sizes = (lens[px] + 1 + lens[py] + 1).astype(np.uint64)
    rp_off = np.concatenate([[0], np.cumsum(rp_sizes)])[:-1].astype(np.uint64)
    rp_total = int(rp_sizes.sum())
    plan = capi.PairhmmPlan()
    plan_fn, launch_fn = (capi.pairhmm5_plan, capi.pairhmm5_launch) if contra else (capi.pairhmm_plan, capi.pairhmm3_launch)
    capi.check(plan_fn(np_local, int(lens[px].max()), int(lens[py].max()), plan))
    pool_cap = int(2 * 24 * np.minimum(lens[px], lens[py]).sum())

    d_codes = torch.from_numpy(codes).to(dev)
    d_tasks = torch.from_numpy(tasks.view(np.int32)).to(dev)
    d_rp_off = torch.from_numpy(rp_off.view(np.int64)).to(dev)
    d_scratch = torch.empty(plan.scratch_bytes // 4, dtype=torch.float32, device=dev)
    # Output buffers come in two sets: with several ranks the all-gather of step k runs on its own stream while the
    # kernel of step k+1 fills the other set.
    force_ex = os.environ.get("DAFS_BENCH_FORCE_EXCHANGE") == "1"  # exercise the exchange path on one rank (tests)
    nsets = 2 if (world > 1 or force_ex) else 1
    sets = []
    for _ in range(nsets):
        o = {"counters": torch.zeros(4, dtype=torch.int64, device=dev),  # [pool_top, queue, status, -]
             "rowptr": torch.empty(rp_total, dtype=torch.int32, device=dev),
             "col": torch.empty(pool_cap, dtype=torch.int32, device=dev),
             "val": torch.empty(pool_cap, dtype=torch.float32, device=dev),
             "pair_off": torch.empty(np_local, dtype=torch.int64, device=dev),
             "pair_nnz": torch.empty(np_local, dtype=torch.int32, device=dev),
             "sim": torch.empty(np_local, dtype=torch.float32, device=dev)}
        a = capi.Pairhmm5Args() if contra else capi.Pairhmm3Args()
        a.codes = d_codes.data_ptr(); a.tasks = d_tasks.data_ptr(); a.ntasks = np_local; a.th = args.th
        a.scratch = d_scratch.data_ptr()
        a.pool_top = o["counters"].data_ptr(); a.queue = o["counters"].data_ptr() + 8; a.status = o["counters"].data_ptr() + 16
        a.rp_off = d_rp_off.data_ptr(); a.rowptr_pool = o["rowptr"].data_ptr()
        a.ent_col = o["col"].data_ptr(); a.ent_val = o["val"].data_ptr(); a.pool_cap = pool_cap
        a.pair_off = o["pair_off"].data_ptr(); a.pair_nnz = o["pair_nnz"].data_ptr(); a.sim = o["sim"].data_ptr()
        (capi.pairhmm5_default_model if contra else capi.pairhmm3_default_model)(C.byref(a.model))
        o["args"] = a
        o["gathered"] = None  # event: the exchange that read this set has finished
        sets.append(o)
    d_counters = sets[0]["counters"]

    stream = torch.cuda.current_stream()
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    ex = None
    comm = None
    if world > 1 or force_ex:
        from dafs_amd import dist as ddist
        ex = ddist.ShardExchange(dist, dev, world, np_local, rp_total, pool_cap)
        comm = torch.cuda.Stream(device=dev)
    sized = [False]
    nstep = [0]

    def step(k=None):
        o = sets[nstep[0] % nsets]
        nstep[0] += 1
        if o["gathered"] is not None:
            stream.wait_event(o["gathered"])  # this set is still being sent
        o["counters"].zero_()
        if k is not None:
            ev0[k].record(stream)
        capi.check(launch_fn(C.byref(o["args"]), C.byref(plan), C.c_void_p(stream.cuda_stream)))
        if k is not None:
            ev1[k].record(stream)
        if ex is not None:
            # the one exchange of the path: every rank ends up with every pair's sparse posteriors.  Payload size:
            # read back once (host sync); the inputs do not change between steps, so later steps reuse the agreed
            # stride.  The gather runs on the communication stream, behind this step's kernel.
            used = None
            if not sized[0]:
                used = int(o["counters"][0].item())
                sized[0] = True
            produced = torch.cuda.Event()
            produced.record(stream)
            comm.wait_event(produced)
            with torch.cuda.stream(comm):
                ex.exchange(o["pair_nnz"], o["sim"], o["pair_off"], o["rowptr"], o["col"], o["val"], used)
                done = torch.cuda.Event()
                done.record(comm)
            o["gathered"] = done

    def drain():
        if comm is not None:
            stream.wait_stream(comm)

    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if int(d_counters[2].item()) != 0:
        raise SystemExit("pair-HMM kernel reported status %d" % int(d_counters[2].item()))

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    drain()  # the last exchange belongs to the timed region
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    kern_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in zip(ev0, ev1)]))
    # algorithmic bytes: ProbCons 28*(L1+1)*(L2+1) per pair = fwd 12C + bwd 12C + posterior 4C;
    # CONTRAlign 44*C = (5 + 5) tables * 4C + posterior 4C
    alg_bytes = float(((44 if contra else 28) * (lens[px] + 1) * (lens[py] + 1)).sum())
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9

    # HBM traffic of the dominant kernel and its secondary (binding) limit: from the committed rocprofv3 PMC passes
    # (profiles/*_pmc.json written by tools/pmc_summary.py, collected and corrected as MI355X_MICROARCH.md prescribes);
    # only quoted when the newest such file is for this very kernel instance and workload, null otherwise.
    traffic = None
    secondary = None
    kname = "k_pairhmm5" if contra else "k_pairhmm3"
    kernel_name = "%s<G=%d,W=%d>" % (kname, plan.group, plan.width)
    pmf = _pmc_file(kernel_name) if (world == 1 and n_seq == cfg_n and args.length == cfg_l) else None
    if pmf is not None and pmf[1].get("config") == args.config:
        pm = pmf[1]
        traffic = pm.get("hbm_bytes_per_launch")
        valu = pm.get("per_launch", {}).get("SQ_INSTS_VALU")
        if valu:
            # a SIMD issues one VALU wave-instruction per 4 cycles at best (16 lanes per clock, 64-lane wavefronts)
            secondary = {"bound": "valu-issue", "valu_wave_instructions_per_launch": valu,
                         "frac": valu * 4.0 / (SIMDS * CLOCK_HZ * kern_ms * 1e-3),
                         "formula": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x 2.4 GHz x kernel_ms)", "pmc_file": "profiles/" + pmf[0]}

    # ---- the timed call path against the oracle: a strided sample of the last launch's outputs, bit for bit ----
    verified = 0
    if rank == 0 and not args.no_verify:
        import oracle_lib
        orc = oracle_lib.load_oracle()
        o = sets[(nstep[0] - 1) % nsets]
        h_nnz = o["pair_nnz"].cpu().numpy().view(np.uint32)
        h_off = o["pair_off"].cpu().numpy().view(np.uint64)
        h_sim = o["sim"].cpu().numpy()
        h_rp = o["rowptr"].cpu().numpy().view(np.uint32)
        for k in range(0, np_local, max(1, np_local // 48)):
            x, y = int(px[k]), int(py[k])
            l1 = int(lens[x])
            n, base = int(h_nnz[k]), int(h_off[k])
            col = o["col"][base:base + n].cpu().numpy().view(np.uint32)
            val = o["val"][base:base + n].cpu().numpy()
            rp = h_rp[int(rp_off[k]):int(rp_off[k]) + l1 + 1]
            orp, ocol, oval = orc.align_calculate(seqs[x], seqs[y], args.th, 1 if contra else 0)
            osim = orc.similarity(orp, ocol, oval, l1, int(lens[y]))
            if not (np.array_equal(rp, orp) and np.array_equal(col, ocol) and val.tobytes() == oval.tobytes()
                    and np.float32(h_sim[k]).tobytes() == np.float32(osim).tobytes()):
                raise SystemExit("bench: pair (%d, %d) of the timed launch differs from the oracle" % (x, y))
            verified += 1

    # ---- the exchange (RCCL all-gather of the packed slabs) against what this rank produced: every pair of this rank's shard
    # must come back from the gathered slab exactly as the kernel wrote it (with one rank and DAFS_BENCH_FORCE_EXCHANGE=1 this
    # is the whole set: the test that runs the nccl = RCCL backend on a one-GPU box)
    exchange_verified = None
    if ex is not None:
        o = sets[(nstep[0] - 1) % nsets]
        g = ex.gathered(lens)
        h_nnz = o["pair_nnz"].cpu().numpy().view(np.uint32)
        h_off = o["pair_off"].cpu().numpy().view(np.uint64)
        h_rp = o["rowptr"].cpu().numpy().view(np.uint32)
        h_col = o["col"].cpu().numpy().view(np.uint32)
        h_val = o["val"].cpu().numpy()
        h_sim = o["sim"].cpu().numpy()
        exchange_verified = 0
        for k in range(0, np_local, max(1, np_local // 256)):
            x, y = int(px[k]), int(py[k])
            n, base, l1, l2 = int(h_nnz[k]), int(h_off[k]), int(lens[x]) + 1, int(lens[y]) + 1
            r0 = int(rp_off[k])
            for tr in (False, True):
                grp, gcol, gval = g.csr(y, x) if tr else g.csr(x, y)
                rp = h_rp[r0 + l1:r0 + l1 + l2] if tr else h_rp[r0:r0 + l1]
                e0 = base + (n if tr else 0)
                if not (np.array_equal(grp, rp) and np.array_equal(gcol, h_col[e0:e0 + n]) and gval.tobytes() == h_val[e0:e0 + n].tobytes()):
                    raise SystemExit("bench: pair (%d, %d) differs after the exchange" % (x, y))
            if np.float32(g.sim(x, y)).tobytes() != np.float32(h_sim[k]).tobytes():
                raise SystemExit("bench: similarity of pair (%d, %d) differs after the exchange" % (x, y))
            exchange_verified += 1

    # ---- the whole sharded phase 1 (north_star: pair jobs AND the consistency transform shard by pair index): folds by
    # x mod G, pair posteriors and the matching transform by pair-index range, three all-gathers of device slabs over RCCL
    # (dist.phase1_sharded), the base-pairing transform replicated.  Timed like the step: barrier + synchronize on both
    # sides, max over ranks.  Not part of `value` (the metric is the pair-posterior path); reported beside it.
    sharded_phase1 = None
    if world > 1 and not args.no_e2e:
        try:
            from dafs_amd import dist as ddist
            sctx = capi.Context(local_rank)
            model = capi.ALIGN_CONTRALIGN if contra else capi.ALIGN_PROBCONS
            reps = []
            for rep in range(3):  # the first builds the buffers
                dist.barrier(); torch.cuda.synchronize()
                t0p = time.perf_counter()
                ddist.phase1_sharded(sctx, seqs, dist, dev, model, args.th, 0.25, 0.25)
                torch.cuda.synchronize(); dist.barrier()
                tt = torch.tensor([time.perf_counter() - t0p], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                reps.append(float(tt.item()))
            sharded_phase1 = {"ms": min(reps[1:]) * 1e3, "first_ms": reps[0] * 1e3, "ranks": world,
                              "what": "folds (x mod G) + pair posteriors and matching consistency transform (pair-index ranges) + 3 all-gathers of "
                                      "device slabs + replicated base-pairing transform; every rank ends with the complete stores"}
            sctx.close()
        except Exception as e:  # noqa: BLE001  (an extra leg must not take the metric line down with it)
            sharded_phase1 = {"error": repr(e)[:300]}

    e2e = None
    dd_forced = None
    stages = None
    if rank == 0 and world == 1 and not args.no_e2e:
        # BASELINE.json's second quantity: end-to-end wall-clock of the whole run (fold, pair posteriors,
        # consistency, tree, progressive DD, final structure) on the same set; not part of `value`.
        from dafs_amd import pipeline
        model = capi.ALIGN_CONTRALIGN if contra else capi.ALIGN_PROBCONS
        ctx = capi.Context(local_rank)
        # skip_uncoupled_folds=False: every node runs all three subproblems of every iteration, as the reference does
        # (the drivers' default leaves out the folding DPs of nodes that no consensus base pair couples; same output)
        pipeline.run(names, seqs, ctx=ctx, align_model=model, skip_uncoupled_folds=False)  # warm-up on the same set: device buffers at their final size, code objects loaded
        t0 = time.perf_counter()
        res = pipeline.run(names, seqs, ctx=ctx, align_model=model, skip_uncoupled_folds=False)
        wall = time.perf_counter() - t0
        its = [v[0] for v in res.dd_log.values()]
        e2e = {"wall_s": wall, "note": "second run on a warm context through the Python driver (buffers allocated, kernels loaded)",
               "flags": "-a %s -s CONTRAfold --no-alifold (defaults otherwise)" % ("CONTRAlign" if contra else "ProbCons"),
               "phases_s": {k: round(v, 4) for k, v in res.seconds.items()},
               "dd_iterations_total": int(np.sum(its)), "dd_iterations_max": int(np.max(its)), "tree_levels": res.levels,
               "columns": len(res.rows[0])}
        if cli is not None and "wall_s" in cli:
            e2e["cold_cli_wall_s"] = cli["wall_s"]
            e2e["cold_cli"] = cli["command"] + ": one cold process (start-up, GPU context, code objects, allocations)"
            e2e["cold_cli_output_equals_driver"] = cli["stdout"] == res.output
        elif cli is not None:
            e2e["cold_cli_error"] = cli.get("error")
        if cpu_out is not None:
            e2e["output_equals_cpu_port"] = cpu_out == res.output
        # BASELINE.json config 3 ("600 subgradient iters"): every node runs t_max iterations (the violated == 0 exit is
        # ignored), so the subproblem kernels are timed on a fixed amount of work
        fr = pipeline.run(names, seqs, ctx=ctx, align_model=model, skip_uncoupled_folds=False, force_iters=1)
        nit = int(np.sum([v[0] for v in fr.dd_log.values()]))
        alg = 0.0
        for i, (l1, l2) in fr.dd_dims.items():
            alg += fr.dd_log[i][0] * (16.0 * (l1 * (l1 - 1) // 2 + l2 * (l2 - 1) // 2) + 13.0 * (l1 + 1) * (l2 + 1) + 64.0 * fr.dd_log[i][2])
        sec = fr.seconds["progressive"]
        dd_forced = {"node_iterations": nit, "nodes": len(fr.dd_log), "progressive_s": sec, "node_iterations_per_s": nit / sec,
                     "algorithmic_bytes": alg, "achieved_GBps": alg / sec / 1e9, "frac_of_hbm_peak": alg / sec / 1e9 / HBM_PEAK_GBS,
                     "note": "k_dd_solve, all nodes forced to t_max=600 iterations; per node-iteration 2*16*L(L-1)/2 + 13*(L1+1)(L2+1) + 64*#cbp "
                             "algorithmic bytes, none of which crosses HBM for narrow nodes: the loop is latency-bound (one wavefront per "
                             "subproblem), the guide tree serialises the nodes"}
        # per-stage device times of one more whole run (HIP events around every launch of the library: kept out of the
        # timed run above), and of one node alone on the device
        ctx.stage_timing(True)
        pipeline.run(names, seqs, ctx=ctx, align_model=model, skip_uncoupled_folds=False)
        rep = ctx.stage_report()
        iso = None
        try:
            one = lambda i: (np.array([i], np.uint32), np.ones((1, int(lens[i])), np.uint8))
            o = ctx.solve_nodes([one(0) + one(1)], capi.dd_params(force_iters=1))[0]
            iso = (ctx.stage_report(), (int(o["iterations"]), int(lens[0]), int(lens[1]), int(o["ncbp"])))
        except capi.DafsHipError:
            iso = None
        # the phase-1 kernels once more, each alone on the device (inside a whole run the folding kernel runs beside the pair
        # and consistency kernels and shares the CUs with them): the numbers a per-kernel roofline should be read from
        alone = {}
        try:
            ctx.set_sequences(seqs)
            for call in (lambda: ctx.fold_posteriors(0.01), lambda: ctx.align_posteriors(capi.ALIGN_CONTRALIGN, args.th, fetch=False),
                         lambda: ctx.align_posteriors(capi.ALIGN_PROBCONS, args.th, fetch=False), lambda: ctx.consistency_match(0.25),
                         lambda: ctx.consistency_bp(0.25)):
                call()              # once untimed: the device has been idle while the host prepared this leg
                ctx.stage_report()
                call()
                alone.update(ctx.stage_report())
        except capi.DafsHipError:
            alone = {}
        ctx.stage_timing(False)
        stages = stage_objects(ctx, seqs, lens, contra, rep, iso)
        if alone:
            solo = stage_objects(ctx, seqs, lens, contra, alone, None)
            stages["alone"] = {k.replace(" (inside the whole run)", ""): {a: b for a, b in v.items() if a in ("ms", "launches", "algorithmic_bytes", "achieved_GBps", "frac", "frac_of_valu_peak")}
                               for k, v in solo.items()}
            stages["alone"]["note"] = "each kernel by itself on the same set (ProbCons store for the transforms); k_pairhmm5 = the CONTRAlign pair kernel, 44*(L1+1)(L2+1) B per pair"
        ctx.close()

    if rank == 0:
        out = {
            "metric": "seq-pairs/sec (all-pairs %s pair-HMM posteriors + sparse rows + sim)" % ("CONTRAlign" if contra else "ProbCons"),
            "value": total_pairs * args.steps / dt,
            "unit": "seq-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "N=%d L~%d synthetic random RNA (seed 12345), %d pairs%s" %
                                   (n_seq, args.length, total_pairs,
                                    "" if world == 1 else ", dealt by cost to %d ranks + 1 all-gather" % world),
                       "align_model": "CONTRAlign" if contra else "ProbCons", "th": args.th,
                       "kernel": "%s<G=%d,W=%d> x %d waves" % (kname, plan.group, plan.width, plan.nwaves)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes, "secondary": secondary},
            "cpu_baseline": cpu,
            "verified_pairs": verified,
        }
        if exchange_verified is not None:
            out["exchange"] = {"backend": dist.get_backend(), "world": world, "verified_pairs": exchange_verified,
                               "slab_bytes_per_rank": int(ex.send.numel()) * 4}
        if e2e is not None:
            out["end_to_end"] = e2e
            out["stages"] = stages
        if dd_forced is not None:
            out["dd_forced_iterations"] = dd_forced
        if sharded_phase1 is not None:
            out["sharded_phase1"] = sharded_phase1
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
