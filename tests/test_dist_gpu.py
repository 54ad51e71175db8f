"""The sharded run on real hardware: two (and three) ranks, one process each, all on the one GPU of the box (gloo
carries the gathers; on a node the same code runs one rank per GPU over RCCL).  Phase 1 is split -- folds by x mod G,
pair posteriors and the matching consistency transform by pair-index range -- gathered and installed, the rest of
the run is replicated.  Every rank must end with the output of the single-process run, bit for bit."""
import os
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, model, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from dafs_amd import capi, pipeline, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        recs = synth.family_set(7, 60, seed=41) + synth.random_set(4, 50, seed=42)
        names, seqs = [r[0] for r in recs], [r[1] for r in recs]
        ctx = capi.Context(0)
        res = pipeline.run(names, seqs, ctx=ctx, align_model=model, shard=(dist, torch.device("cuda", 0)))
        sim = ctx.sim()
        ctx.close()
        q.put((rank, res.output, sim.tobytes()))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "FAILED " + traceback.format_exc(), b""))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,model", [(2, 0), (3, 1)])
def test_sharded_run_equals_single_process(world, model):
    import torch.multiprocessing as mp
    from dafs_amd import capi, pipeline, synth
    recs = synth.family_set(7, 60, seed=41) + synth.random_set(4, 50, seed=42)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    ctx = capi.Context(0)
    want = pipeline.run(names, seqs, ctx=ctx, align_model=model)
    want_sim = ctx.sim().tobytes()
    ctx.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, world, port, model, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(r[0] for r in res) == list(range(world))
    for rank, out, sim in res:
        assert not out.startswith("FAILED"), out
        assert out == want.output, rank
        assert sim == want_sim, rank


def test_rccl_exchange_on_one_rank():
    """The packed-slab all-gather of bench.py's multi-GPU step through the nccl backend (= RCCL on ROCm), world size 1 on the
    box's one GPU: DAFS_BENCH_FORCE_EXCHANGE=1 makes the single rank run the exchange path (two output sets, gather on a
    communication stream), and bench.py checks every sampled pair of the gathered slab against the kernel's own output."""
    import json
    import subprocess
    env = dict(os.environ, DAFS_BENCH_FORCE_EXCHANGE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--config", "c2",
                        "--no-cpu", "--no-e2e"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["exchange"]["backend"] == "nccl" and line["exchange"]["world"] == 1
    assert line["exchange"]["verified_pairs"] >= 200 and line["verified_pairs"] > 0


def _worker_nccl(q, port, host_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from dafs_amd import capi, pipeline, synth
    from dafs_amd import dist as dd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        recs = synth.family_set(7, 60, seed=41) + synth.random_set(4, 50, seed=42)
        names, seqs = [r[0] for r in recs], [r[1] for r in recs]
        if host_path:
            dd.phase1_sharded, keep = dd.phase1_sharded_host, dd.phase1_sharded
        ctx = capi.Context(0)
        res = pipeline.run(names, seqs, ctx=ctx, align_model=1, shard=(dist, dev))
        sim = ctx.sim()
        ctx.close()
        q.put((res.output, sim.tobytes(), dist.get_backend()))
    except Exception:  # noqa: BLE001
        import traceback
        q.put(("FAILED " + traceback.format_exc(), b"", ""))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("host_path", [False, True])
def test_sharded_whole_run_over_rccl(host_path):
    """The whole sharded run with the nccl backend (= RCCL), world size 1 on the box's one GPU: phase 1's three exchanges are
    all_gather_into_tensor calls on DEVICE buffers (dist.gather_parts; dafs_hip_*_export_dev / *_install_dev on either
    side), nothing of the stores passes through host memory.  Must equal the ordinary single-process run; the host-array
    form of round 2 (phase1_sharded_host) is run through the same backend as the cross-check."""
    import torch.multiprocessing as mp
    from dafs_amd import capi, pipeline, synth
    recs = synth.family_set(7, 60, seed=41) + synth.random_set(4, 50, seed=42)
    names, seqs = [r[0] for r in recs], [r[1] for r in recs]
    ctx = capi.Context(0)
    want = pipeline.run(names, seqs, ctx=ctx, align_model=1)
    want_sim = ctx.sim().tobytes()
    ctx.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    p = mpc.Process(target=_worker_nccl, args=(q, _free_port(), host_path))
    p.start()
    out, sim, backend = q.get(timeout=120)
    p.join(timeout=60)
    assert not out.startswith("FAILED"), out
    assert backend == "nccl" and out == want.output and sim == want_sim
