"""Worker functions for the multi-process tests (importable by torch.multiprocessing.spawn)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    return dist


def hook_on_host_memory(rank, world, port, outdir):
    """GlooAllReduce(host=True): the all-reduce hook contract on plain host buffers."""
    dist = _init(rank, world, port)
    from mcptam_amd.dist import GlooAllReduce
    h = GlooAllReduce(host=True)
    a = np.arange(1000, dtype=np.float64) * (rank + 1)
    h(a.ctypes.data, a.size, 0)
    b = np.array([float(rank == r) for r in range(world)])
    h(b.ctypes.data, b.size, 0)
    np.savez(os.path.join(outdir, "host_%d.npz" % rank), a=a, b=b, calls=h.calls)
    dist.destroy_process_group()


def sharded_solve_on_one_gpu(rank, world, port, outdir, cfg, iters):
    """Every rank owns one shard of the map; poses replicated; reduced system summed through the hook."""
    dist = _init(rank, world, port)
    from mcptam_amd import chain_bundle, synth
    from mcptam_amd.dist import GlooAllReduce
    from helpers import collect
    p = synth.make_config(shard=rank, **cfg)
    b = chain_bundle.ChainBundle(p.cams, True, True, False, device=0)
    ids = p.populate(b)
    hook = GlooAllReduce(host=False)
    b.SetAllReduce(hook, rank, world)
    rc = b.Compute(iters)
    R, t, X = collect(b, ids)
    logs = np.array([[l["chi2_start"], l["chi2_end"], l["lambda_end"], l["sigma_sq"], l["trials"], l["accepted"]] for l in b.IterLogs()])
    tm = b.Timing()
    np.savez(os.path.join(outdir, "shard_%d.npz" % rank), rc=rc, R=R, t=t, X=X, logs=logs, sigma_sq=b.GetSigmaSquared(),
             mean_chi2=b.GetMeanChiSquared(), calls=hook.calls, n_out=len(b.GetOutlierMeasurements()),
             coll_main=tm["n_collectives_main"], coll_spec=tm["n_collectives_spec"], median_fast=tm["n_median_fast"],
             trials=tm["n_trials"], solves=tm["n_solves"], bytes_main=tm["collective_bytes_main"], bytes_spec=tm["collective_bytes_spec"])
    b.close()
    dist.destroy_process_group()


def partitioned_solve_on_one_gpu(rank, world, port, outdir, cfg, iters):
    """ONE map split by synth.partition (points by source MKF, measurement counts balanced): this rank adjusts its block."""
    dist = _init(rank, world, port)
    from mcptam_amd import chain_bundle, synth
    from mcptam_amd.dist import GlooAllReduce
    from helpers import collect
    p = synth.partition(synth.make_config(**cfg), world, rank)
    b = chain_bundle.ChainBundle(p.cams, True, True, False, device=0)
    ids = p.populate(b)
    b.SetAllReduce(GlooAllReduce(host=False), rank, world)
    rc = b.Compute(iters)
    R, t, X = collect(b, ids)
    logs = np.array([[l["chi2_start"], l["chi2_end"], l["lambda_end"], l["sigma_sq"], l["trials"], l["accepted"]] for l in b.IterLogs()])
    np.savez(os.path.join(outdir, "part_%d.npz" % rank), rc=rc, R=R, t=t, X=X, logs=logs, points=p.part["points"], sigma_sq=b.GetSigmaSquared(),
             n_out=len(b.GetOutlierMeasurements()), n_meas=p.n_meas)
    b.close()
    dist.destroy_process_group()


def _empty_copy(p):
    """The same trajectory and cameras with no point and no measurement (a rank whose shard of the map is empty)."""
    import copy
    q = copy.copy(p)
    for name in ("pt_x", "pt_src", "pt_fixed", "ms_mkf", "ms_cam", "ms_pt", "ms_uv", "ms_level"):
        setattr(q, name, getattr(p, name)[:0])
    q.true_world = None
    return q


def sharded_solve_with_an_empty_rank(rank, world, port, outdir, cfg, iters):
    """Rank 0 holds the whole map, rank 1 nothing: rank 1 must still join every collective (no hang) and end with the same poses."""
    dist = _init(rank, world, port)
    from mcptam_amd import chain_bundle, synth
    from mcptam_amd.dist import GlooAllReduce
    p = synth.make_config(shard=0, **cfg)
    if rank == 1:
        p = _empty_copy(p)
    b = chain_bundle.ChainBundle(p.cams, True, True, False, device=0)
    ids = p.populate(b)
    b.SetAllReduce(GlooAllReduce(host=False), rank, world)
    rc = b.Compute(iters)
    R, t = b.GetPoses(ids["mkf"])
    np.savez(os.path.join(outdir, "empty_%d.npz" % rank), rc=rc, R=R, t=t, sigma_sq=b.GetSigmaSquared(), max_cov=b.GetMaxCov(),
             trials=np.array([l["trials"] for l in b.IterLogs()]), n_out=len(b.GetOutlierMeasurements()))
    b.close()
    dist.destroy_process_group()


def sharded_max_cov(rank, world, port, outdir, cfg, iters):
    """Fewer than three free poses, points sharded over two ranks: GetMaxCov is the GLOBAL median of the depth covariances."""
    dist = _init(rank, world, port)
    from mcptam_amd import chain_bundle, synth
    from mcptam_amd.dist import GlooAllReduce
    p = synth.make_config(shard=rank, **cfg)
    b = chain_bundle.ChainBundle(p.cams, True, True, False, device=0)
    ids = p.populate(b)
    b.SetAllReduce(GlooAllReduce(host=False), rank, world)
    rc = b.Compute(iters)
    R, t = b.GetPoses(ids["mkf"])
    np.savez(os.path.join(outdir, "cov_%d.npz" % rank), rc=rc, R=R, t=t, max_cov=b.GetMaxCov())
    b.close()
    dist.destroy_process_group()


def rccl_hook_single_rank(rank, world, port, outdir):
    """RcclAllReduce on a device buffer with a 1-rank nccl (= RCCL) group: staging copies, stream sync, collective."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from mcptam_amd.dist import RcclAllReduce
    h = RcclAllReduce(torch.device("cuda", 0))
    t = torch.arange(5000, dtype=torch.float64, device="cuda") * 0.5
    h(t.data_ptr(), t.numel(), 0)
    h(t.data_ptr() + 8 * 100, 50, 0)
    torch.cuda.synchronize()
    np.savez(os.path.join(outdir, "rccl.npz"), t=t.cpu().numpy(), calls=h.calls)
    dist.destroy_process_group()


def forced_multi_native_rccl(rank, world, port, outdir, cfg, iters):
    """MCP_BA_FORCE_MULTI=1 with a one-rank RCCL communicator: the two-lane multi-rank machine (packed-tile all-reduces on both
    streams, trial blocks with riding histograms, one-collective medians) on the real transport.  Sums over one rank are exact."""
    os.environ["MCP_BA_FORCE_MULTI"] = "1"
    import torch
    torch.cuda.set_device(0)
    from mcptam_amd import chain_bundle, synth
    from helpers import collect
    comm = chain_bundle.Comm(chain_bundle.comm_unique_id(), 0, 1, 0)
    t = torch.arange(4096, dtype=torch.float64, device="cuda") * 0.25
    comm.allreduce(t.data_ptr(), t.numel(), lane=1)
    p = synth.make_config(**cfg)
    b = chain_bundle.ChainBundle(p.cams, True, True, False, disable_convergence=True, device=0)
    ids = p.populate(b)
    b.SetComm(comm)
    rc = b.Compute(iters)
    R, tt, X = collect(b, ids)
    tm = b.Timing()
    logs = np.array([[l["chi2_start"], l["chi2_end"], l["lambda_end"], l["sigma_sq"], l["trials"], l["accepted"]] for l in b.IterLogs()])
    np.savez(os.path.join(outdir, "forced.npz"), t=t.cpu().numpy(), rc=rc, R=R, tt=tt, X=X, logs=logs, sigma_sq=b.GetSigmaSquared(),
             coll_main=tm["n_collectives_main"], coll_spec=tm["n_collectives_spec"], median_fast=tm["n_median_fast"], trials=tm["n_trials"], solves=tm["n_solves"])
    b.close(); comm.close()


def watchdog_stall(rank, world, port, outdir):
    """The all-reduce hook parks a long sleep kernel on the solver's stream once: the next bounded wait must give up."""
    os.environ["MCP_BA_FORCE_MULTI"] = "1"
    os.environ["MCP_BA_TIMEOUT_MS"] = "150"
    import time
    import torch
    torch.cuda.set_device(0)
    from mcptam_amd import chain_bundle, synth
    p = synth.make_config("c1")
    g = chain_bundle.ChainBundle(p.cams, True, True, False, disable_convergence=True, device=0)
    p.populate(g)
    n = [0]

    def stalling(ptr, count, stream):
        n[0] += 1
        if n[0] == 12 and stream:
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=torch.device("cuda", 0))):
                torch.cuda._sleep(int(6e9))
    g.SetAllReduce(stalling, 0, 1)
    t0 = time.time()
    msg = "no error"
    try:
        g.Compute(10)
    except RuntimeError as exc:
        msg = str(exc)
    dt = time.time() - t0
    torch.cuda.synchronize()
    g.close()
    np.savez(os.path.join(outdir, "watchdog.npz"), msg=msg, seconds=dt, calls=n[0])


def native_rccl_single_rank(rank, world, port, outdir):
    """The library's own RCCL communicator (dlopen'ed librccl, stream-ordered ncclAllReduce) with one rank:
    bootstrap through torch.distributed, raw all-reduce, and a full solve with the communicator installed."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group(backend="gloo", rank=0, world_size=1)
    from mcptam_amd import chain_bundle, synth
    from mcptam_amd.dist import init_rccl_comm
    from helpers import collect
    comm = init_rccl_comm(0, 1, 0)
    t = torch.arange(4096, dtype=torch.float64, device="cuda") * 0.25
    comm.allreduce(t.data_ptr(), t.numel())
    p = synth.make_config("tiny")
    b = chain_bundle.ChainBundle(p.cams, True, True, False, device=0)
    ids = p.populate(b)
    b.SetComm(comm)
    rc = b.Compute(8)
    R, tt, X = collect(b, ids)
    np.savez(os.path.join(outdir, "native.npz"), t=t.cpu().numpy(), rc=rc, R=R, tt=tt, X=X)
    b.close(); comm.close()
    dist.destroy_process_group()


def sharded_pose_refine(rank, world, port, outdir):
    """One camera per rank: every rank refines the base pose from its own camera's points; the Tukey median and the 6x6 + 6
    accumulator are exchanged per iteration (mcp_track_pose_refine_sharded)."""
    dist = _init(rank, world, port)
    import test_oracle_cpu as toc
    from mcptam_amd.dist import GlooAllReduce
    from mcptam_amd.keyframe import track_pose_refine_sharded
    cam, cfbs, bfw, recs = toc._refine_scene()
    mine = recs[recs["cam"] == rank]
    cap = int(max((recs["cam"] == r).sum() for r in range(world)))
    hook = GlooAllReduce(host=False)
    pose, mu, w, out = track_pose_refine_sharded(mine, [cam, cam], cfbs, bfw, allreduce=hook, rank=rank, world=world, cap=cap)
    np.savez(os.path.join(outdir, "refine_%d.npz" % rank), R=pose[0], t=pose[1], mu=mu, w=w, image=out["image"], calls=hook.calls)
    dist.destroy_process_group()
