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

With N > 1 the pairs of a sqrt(N)-times larger sequence set are dealt to the ranks by cost
(weak scaling: ~8128 pairs per GPU), each rank runs its shard, and the sparse posteriors are
exchanged with one all-gather over RCCL (the only collective of the path).

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
BASE_N, BASE_L = 128, 150


def build_shard(n_seq, length, world, rank, seed=12345):
    from dafs_amd import dist as ddist, synth
    recs = synth.random_set(n_seq, length, seed=seed)
    names = [n for n, _ in recs]
    seqs = [s for _, s in recs]
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    px, py, total = ddist.shard_pairs(lens, world, rank)
    return names, seqs, lens, px, py, total


def cpu_baseline(seqs, px, py, th, budget_s=15.0):
    """CPU path on a bounded sample of this rank's pairs, 1 thread.
    kind "reference": the reference's own ProbCons::calculate (oracle/_ref, built from
    /root/reference by oracle/Makefile) for the posterior + sparse rows, plus the oracle's
    restatement of transpose_mp / calculate_similarity_score (dafs.cpp is not compilable here).
    kind "port": the oracle restatement for everything."""
    import oracle_lib
    orc = oracle_lib.load_oracle()
    ref = oracle_lib.load_ref()
    kind = "reference" if ref is not None else "port"
    calc = ref.align_calculate if ref is not None else orc.align_calculate
    n = 0
    t0 = time.perf_counter()
    step = 61  # stride through the cost-sorted list so any prefix of the sample spans all lengths
    for k in [k for r in range(step) for k in range(r, len(px), step)]:
        a, b = seqs[px[k]], seqs[py[k]]
        rp, col, val = calc(a, b, th)
        orc.similarity(rp, col, val, len(a), len(b))
        # transpose_mp
        rows = np.repeat(np.arange(len(a), dtype=np.uint32), np.diff(rp))
        o = np.lexsort((rows, col))
        _ = rows[o], val[o]
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "seq-pairs/s", "cores": 1, "kind": kind,
            "sample": "%d of this rank's %d pairs (N=%d L~%d set), posterior+sparse rows+transpose+sim, %.1f s" %
                      (n, len(px), len(seqs), BASE_L, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-seq", type=int, default=0, help="override the number of sequences")
    ap.add_argument("--length", type=int, default=BASE_L)
    ap.add_argument("--th", type=float, default=0.01)
    ap.add_argument("--model", choices=("probcons", "contralign"), default="probcons", help="alignment model of the timed kernel")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end wall-clock leg")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from dafs_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): launch with torch.distributed.run" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or os.environ.get("DAFS_BENCH_FORCE_EXCHANGE") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if world == 1:
            dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
        else:
            dist.init_process_group("nccl", device_id=dev)

    n_seq = args.n_seq or int(round(BASE_N * math.sqrt(world)))
    names, seqs, lens, px, py, total_pairs = build_shard(n_seq, args.length, world, rank)
    np_local = len(px)

    # ---- device-resident inputs (torch = allocator + stream only) ----
    codes = np.concatenate([capi.encode(s) for s in seqs])
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    tasks = np.zeros((np_local, 4), np.uint32)
    tasks[:, 0] = off[px]; tasks[:, 1] = lens[px]; tasks[:, 2] = off[py]; tasks[:, 3] = lens[py]
    rp_sizes = (lens[px] + 1 + lens[py] + 1).astype(np.uint64)
    rp_off = np.concatenate([[0], np.cumsum(rp_sizes)])[:-1].astype(np.uint64)
    rp_total = int(rp_sizes.sum())
    plan = capi.PairhmmPlan()
    contra = args.model == "contralign"
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

    # HBM traffic of the dominant kernel: from the committed rocprofv3 PMC passes (profiles/*_pmc.json,
    # collected and corrected as MI355X_MICROARCH.md prescribes); only quoted when it is the same kernel.
    traffic = None
    kname = "k_pairhmm5" if contra else "k_pairhmm3"
    kernel_name = "%s<G=%d,W=%d" % (kname, plan.group, plan.width)
    try:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json"))):
            pm = json.load(open(f))
            if pm.get("kernel", "").startswith(kernel_name) and world == 1 and n_seq == BASE_N and args.length == BASE_L:
                traffic = pm["hbm_bytes_per_launch"]
    except Exception:  # noqa: BLE001
        traffic = None

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(seqs, px, py, args.th)

    e2e = None
    if rank == 0 and world == 1 and not args.no_e2e:
        # BASELINE.json's second quantity: end-to-end wall-clock of the whole run (fold, pair posteriors,
        # consistency, tree, progressive DD, final structure) on the same set; not part of `value`.
        from dafs_amd import pipeline
        ctx = capi.Context(local_rank)
        # skip_uncoupled_folds=False: every node runs all three subproblems of every iteration, as the reference does
        # (the drivers' default leaves out the folding DPs of nodes that no consensus base pair couples; same output)
        pipeline.run(names, seqs, ctx=ctx, skip_uncoupled_folds=False)  # warm-up on the same set: device buffers at their final size, code objects loaded
        t0 = time.perf_counter()
        res = pipeline.run(names, seqs, ctx=ctx, skip_uncoupled_folds=False)
        wall = time.perf_counter() - t0
        its = [v[0] for v in res.dd_log.values()]
        e2e = {"wall_s": wall, "note": "second run on a warm context (buffers allocated, kernels loaded)", "flags": "-a ProbCons -s CONTRAfold --no-alifold (defaults otherwise)",
               "phases_s": {k: round(v, 4) for k, v in res.seconds.items()},
               "dd_iterations_total": int(np.sum(its)), "dd_iterations_max": int(np.max(its)), "tree_levels": res.levels,
               "columns": len(res.rows[0])}
        ctx.close()

    if rank == 0:
        out = {
            "metric": "seq-pairs/sec (all-pairs %s pair-HMM posteriors + sparse rows + sim)" % ("CONTRAlign" if contra else "ProbCons"),
            "value": total_pairs * args.steps / dt,
            "unit": "seq-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "N=%d L~%d synthetic random RNA (seed 12345), %d pairs%s" %
                                   (n_seq, args.length, total_pairs,
                                    "" if world == 1 else ", dealt by cost to %d ranks + 1 all-gather" % world),
                       "align_model": "CONTRAlign" if contra else "ProbCons", "th": args.th,
                       "kernel": "%s<G=%d,W=%d> x %d waves" % (kname, plan.group, plan.width, plan.nwaves)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": alg_bytes},
            "cpu_baseline": cpu,
        }
        if e2e is not None:
            out["end_to_end"] = e2e
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
