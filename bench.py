#!/usr/bin/env python
"""bench.py -- ChainBundle LM iterations/s on the BASELINE.json metric map (MI355X).

A "step" is one LM outer iteration (one linearisation + T>=1 trial solves + evaluations + one
Huber sigma^2 recompute) over one 4-camera, 200-MKF, 50k-point, 400k-measurement
synthetic map shard, with the map resident in HBM before the timed region.  With N ranks every
rank holds all 200 poses and its own 50k-point / 400k-measurement shard (weak scaling); the
reduced pose system is summed with RCCL once per trial.  value = N * K / time.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROUND = "r06"                # profiles/<ROUND>/ holds this round's rocprofv3 summaries; files of other rounds are never read
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_PEAK_TFLOPS = 78.6      # fp64 vector == matrix rate on MI355X (SURVEY.md 8(d) nominal; not in the guide's table)


TRAFFIC_FILE = os.path.join("profiles", ROUND, "pmc_traffic_per_launch.json")


def load_traffic():
    """Per-kernel HBM traffic per launch from THIS round's PMC pass (scripts/gpu_final.sh: two separate `rocprofv3 --pmc`
    runs of this file, parsed by scripts/parse_traffic.py): bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE is
    doubled on gfx950 as MI355X_MICROARCH.md prescribes.  The file records the kernel set it was taken with; if a kernel
    of the current build is missing from it (a stale file) the stage's traffic is reported as null, never guessed."""
    path = os.path.join(ROOT, TRAFFIC_FILE)
    if not os.path.exists(path):
        return {}
    raw = json.load(open(path))
    return {k.replace("void ", ""): (2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0 for k, v in raw.items()}


STAGE_KERNELS = {       # kernels launched per stage invocation: (name, count) ; count -1 = number of 32-column Cholesky steps
    "linearize": [("mcp::k_linearize_pipe", 1)],          # round 5: the software-pipelined measurement loop (k_linearize_group: MCP_BA_LIN_PIPE=0)
    "schur": [("mcp::k_schur4", 1), ("mcp::k_assemble", 1)],      # round 5: every system of the batch in one workgroup per group (ba_schur4.h)
    "backsub_update": [("mcp::k_backsub", 1), ("mcp::k_update_poses", 1)],
    "eval": [("mcp::k_eval<true>", 1), ("mcp::k_chains", 1), ("mcp::k_final_sums", 1)],
    "select": [("mcp::k_select_pass", 2), ("mcp::k_select_gather", 1), ("mcp::k_select_small", 1)],
    "cholesky": [("mcp::k_chol_persist", 1)],        # one persistent launch per solve (ba_chol2.h; k_chol_persist_seg for a plan of several chains); MCP_BA_CHOL_PERSIST=0: k_chol_step x steps
    "tri_solve": [("mcp::k_chol_back2", 1)],
}


def stage_rooflines(tm, M, N, np_, n_lin, n_trials, n_solves):
    """Algorithmic bytes / flops per launch group (meas record 36 B, point 24 B,
    V+g 72 B, chi2 8 B) over the measured per-launch duration of each stage.

    A reduced-system solve (Schur + Cholesky + back-substitution launches) carries the trial's system plus speculative
    ones for the next lambdas of the LM schedule; only the systems a trial actually consumed count as algorithmic work:
    n_trials / n_solves systems per solve on average."""
    n_solves = max(n_solves, 1)
    useful = n_trials / n_solves
    out = {}
    def hbm(name, ms_total, launches, nbytes):
        if launches and ms_total > 0:
            dur = ms_total / launches * 1e-3
            ach = nbytes / dur / 1e9
            out[name] = dict(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                             traffic=None, avg_ms=ms_total / launches, launches=launches, bytes_per_launch=nbytes)
    hbm("linearize", tm["linearize_ms"], n_lin, M * 36 + N * 24 + N * 72)
    hbm("schur", tm["schur_ms"], n_solves, useful * (M * 36 + N * 72))
    if "schur" in out and tm.get("schur_mfma_per_system", 0) > 0:
        # `bound: hbm` keeps SURVEY 8(d)'s algorithmic-byte basis for the fraction, but the stage is not bandwidth-bound: a workgroup is
        # a chain of phases (index hops, W fetch, LDS staging, matrix-core products, flush) and the products themselves are zero-filled
        # 16 x 16 x 4 tiles.  Said here in numbers: what the matrix pipe executes (every system of the batch), what of that is
        # structurally non-zero work, and how busy that keeps the fp64 matrix cores over the stage's duration.
        r = out["schur"]
        dur = r["avg_ms"] * 1e-3
        nsys_built = 4.0
        exec_flops = tm["schur_mfma_per_system"] * 2048.0 * nsys_built
        r["limited_by"] = "latency of one workgroup's phase chain + matrix-core issue (DESIGN.md 4), not HBM"
        r["mfma_executed"] = {"instructions_per_launch": tm["schur_mfma_per_system"] * nsys_built, "tflops": exec_flops / dur / 1e12,
                              "frac_of_fp64_mfma_peak": exec_flops / dur / 1e12 / FP64_PEAK_TFLOPS,
                              "structural_tflops_consumed": useful * tm["schur_flops_structural"] / dur / 1e12,
                              "zero_fill_factor": tm["schur_mfma_per_system"] * 2048.0 / max(tm["schur_flops_structural"], 1.0),
                              "note": "instructions = non-empty 16-row tile pairs of every 16-point chunk x 12 x the 4 systems of a batch; structural = "
                                      "sum over points of k(k+1)/2 x 324 flop (k poses see the point), only the systems a trial consumed"}
    hbm("backsub_update", tm["update_ms"], n_trials, M * 36 + N * 72 + N * 24)
    hbm("eval", tm["eval_ms"], n_trials + n_lin, M * 36 + N * 24 + M * 8)
    hbm("select", tm["select_ms"], n_lin + 1, M * 8)
    for name, key, flops in (("cholesky", "cholesky_ms", np_ ** 3 / 3.0), ("tri_solve", "solve_ms", 2.0 * np_ ** 2)):
        if n_trials and tm[key] > 0:
            dur = tm[key] / n_solves * 1e-3
            ach = useful * flops / dur / 1e12
            out[name] = dict(bound="mfma", achieved=ach, peak=FP64_PEAK_TFLOPS, unit="TFLOP/s", frac=ach / FP64_PEAK_TFLOPS,
                             traffic=None, avg_ms=tm[key] / n_solves, launches=n_solves, flops_per_launch=useful * flops)
            if name == "cholesky" and tm.get("chol_flops_plan", 0) > 0:
                # SURVEY 8(d)'s basis above is the DENSE (6P)^3 / 3; the plan only touches the tiles a banded trajectory fills
                pl = useful * tm["chol_flops_plan"] / dur / 1e12
                out[name]["plan_basis"] = {"flops_per_factorisation": tm["chol_flops_plan"], "dense_flops": flops, "achieved": pl, "frac": pl / FP64_PEAK_TFLOPS,
                                           "note": "tile operations of the block-sparse plan after symbolic fill (mcp_ba_timing.chol_flops_plan); `achieved` / `frac` above keep the dense basis of SURVEY 8(d)"}
    traffic = load_traffic()
    steps = (np_ + 31) // 32
    for name, r in out.items():
        t = 0.0
        ok = bool(traffic)
        for kname, cnt in STAGE_KERNELS.get(name, []):
            if kname == "mcp::k_chol_persist" and tm.get("chol_chains", 1) > 1:
                kname = "mcp::k_chol_persist_seg"          # the plan of several chains is walked by the segment-aware kernel (ba_chol2.h)
            if kname not in traffic:
                ok = False
                break
            t += traffic[kname] * (steps if cnt < 0 else cnt)
        r["traffic"] = t if ok else None
    return out


def usable_cores():
    """Cores this process may actually run on: the affinity mask, capped by the cgroup CPU quota if there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:                                                   # cgroup v2
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    try:                                                   # cgroup v1
        quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if quota > 0 and period > 0:
            n = max(1, min(n, quota // period))
    except Exception:
        pass
    return n


def native_oracle():
    """Builds oracle/ for THIS box (-O3 -march=native -fopenmp) into a scratch directory and points the oracle loader at it;
    falls back to the portable liborc.so (no OpenMP) if the box has no compiler."""
    import subprocess
    import tempfile
    d = tempfile.mkdtemp(prefix="orc_native_")
    try:
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "native", "ORC_NATIVE_DIR=" + d],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        os.environ["ORC_LIB"] = os.path.join(d, "liborc_native.so")
        return "built on this box with gcc -O3 -march=native -fopenmp"
    except Exception as exc:
        return "portable build gcc -O2, no OpenMP (native build failed: %r)" % (exc,)


def parity_against_oracle(problem, chain_bundle, device, gpu_logs, o, oids, n):
    """Iteration log of the timed GPU run against the oracle's over the oracle's n iterations (trial counts, accept/reject,
    chi2, lambda), and the state of a GPU run stopped after the same n iterations against the oracle's state."""
    import numpy as np
    olog = o.IterLogs()
    m = min(n, len(gpu_logs), len(olog))
    flips, dchi, dlam = 0, 0.0, 0.0
    for g, r in zip(gpu_logs[:m], olog[:m]):
        if g["trials"] != r["trials"] or g["accepted"] != r["accepted"]:
            flips += 1
            break
        dchi = max(dchi, abs(g["chi2_start"] - r["chi2_start"]) / abs(r["chi2_start"]), abs(g["chi2_end"] - r["chi2_end"]) / abs(r["chi2_end"]))
        dlam = max(dlam, abs(g["lambda_end"] - r["lambda_end"]) / abs(r["lambda_end"]))
    b = chain_bundle.ChainBundle(problem.cams, True, True, False, disable_convergence=True, device=device)
    ids = problem.populate(b)
    b.Compute(n)
    Rg, tg = b.GetPoses(ids["mkf"])
    Xg = b.GetPoints(ids["point"])
    b.close()
    Ro = np.array([o.GetPose(int(i))[0] for i in oids["mkf"]])
    to = np.array([o.GetPose(int(i))[1] for i in oids["mkf"]])
    Xo = np.array([o.GetPoint(int(i)) for i in oids["point"]])
    rel = lambda a, c: float(np.abs(a - c).max() / np.abs(c).max())
    out = {"iterations_compared": m, "branch_flips": flips, "max_rel_chi2": dchi, "max_rel_lambda": dlam,
           "state_rel_err": {"pose_R": rel(Rg, Ro), "pose_t": rel(tg, to), "points": rel(Xg, Xo)}, "tolerance": 1e-6}
    out["ok"] = bool(flips == 0 and dchi < 1e-7 and dlam < 1e-6 and max(out["state_rel_err"].values()) < 1e-6)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="metric")
    ap.add_argument("--cpu-iters", type=int, default=8, help="LM iterations of the CPU baseline sample (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-tracker", action="store_true", help="skip the secondary Tracker line (BASELINE config c3) that is appended under `tracker_c3`")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: every rank brings its own shard of the configuration's size (the headline); strong: ONE map of the configuration's "
                         "size is partitioned over the ranks (synth.partition: points by source MKF, measurement counts balanced) -- BASELINE c4 as stated")
    ap.add_argument("--debug-single-device", action="store_true",
                    help="all ranks on GPU 0 with a host-staged gloo all-reduce: exercises the multi-rank code path on a 1-GPU box; "
                         "the printed value is NOT a valid measurement (config.debug says so)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched as plain `python bench.py --gpus N`: become the launcher (one process per GPU over RCCL, as the contract says)
        import socket
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                                  "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:])
    # stdout carries ONE line, the result: whatever a library prints there on its own (RCCL's version banner when a communicator is
    # created) goes to stderr instead -- file descriptor 1 is pointed at stderr for the run, the result line is written to the real one
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world))
    if args.debug_single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        if args.debug_single_device:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    from mcptam_amd import chain_bundle, synth
    from mcptam_amd.dist import GlooAllReduce, RcclAllReduce, init_rccl_comm

    if args.scaling == "strong" and world > 1:
        problem = synth.partition(synth.make_config(args.config), world, rank)      # one map, this rank's block of its points
    else:
        problem = synth.make_config(args.config, shard=rank)
    # transport of the per-trial all-reduce: the library's own RCCL communicator on the solver stream; if that cannot
    # be created, the torch.distributed (backend nccl = RCCL) hook
    comm, hook, transport = None, None, "none"
    if world > 1 and args.debug_single_device:
        hook = GlooAllReduce()
        transport = "DEBUG: gloo, host-staged, all ranks on one GPU"
    elif world > 1:
        try:
            comm = init_rccl_comm(rank, world, local_rank)
            transport = "rccl (library communicator, stream-ordered)"
        except Exception as exc:
            if rank == 0:
                print("bench: native RCCL communicator unavailable (%r); using the torch.distributed hook" % (exc,), file=sys.stderr)
        # every rank must use the same transport: fall back together if any rank could not join
        flag = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if comm is not None:
                comm.close()
                comm = None
            hook = RcclAllReduce(dev)
            transport = "rccl via torch.distributed hook"

    setup_ms = {}

    def fresh(profile=False):
        b = chain_bundle.ChainBundle(problem.cams, True, True, False, disable_convergence=True, device=local_rank, profile=profile)
        t0 = time.perf_counter()
        problem.populate(b)
        t1 = time.perf_counter()
        if comm is not None:
            b.SetComm(comm)
        elif hook is not None:
            b.SetAllReduce(hook, rank, world)
        b.Prepare()          # structure + upload: the map is resident in HBM before the timed region
        t2 = time.perf_counter()
        # what a BundleAdjust() call pays once before its first LM iteration (the reference builds a fresh ChainBundle per
        # call, BundleAdjusterMulti.cc:75): populate = the AddPose/AddPoint/AddMeas replay through the C ABI (batched
        # entries, from Python here), prepare = symbolic structure on the host + upload over PCIe
        setup_ms["populate_ms"] = (t1 - t0) * 1e3                  # from Python: numpy marshalling + the library's entries
        setup_ms["populate_in_library_ms"] = b.abi_seconds * 1e3     # what a native caller (shim/ChainBundle.cc) pays: the bulk Add* entries alone
        setup_ms["prepare_ms"] = (t2 - t1) * 1e3
        return b

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warmup: W untimed LM iterations (code objects, allocations, clocks) on a second handle, run AFTER the timed handle has been
    # populated and prepared (60 ms of host work during which the device would idle and clock down) and right before the timed region
    PREWARM = 40     # untimed iterations before the W warm-up ones: brings the device clocks up after an idle or profiled period
    bw = fresh() if args.warmup > 0 else None
    # the set-up figures are medians over a few whole set-ups (fresh handle, replay, Prepare, close), the first one (cold allocator,
    # first touch of the pinned staging) left out -- a single sample of a ~10 ms host phase moves by +-30 %
    # Two figures for Prepare(): COLD -- the structure cache emptied before every sample: a map whose topology the process has not seen
    # (the first adjustment after a keyframe or points were added) -- and CACHED: the topology of the call before, which is what
    # MCPTAM's repeated adjustments of an unconverged map bring (src/MapMaker.cc run loop; include/mcp_ba.h structure cache).
    samples, cold = [], []
    if world == 1 and args.warmup > 0:
        for _ in range(4):
            chain_bundle.struct_cache_clear()
            fresh().close()
            cold.append(dict(setup_ms))
        for _ in range(6):
            fresh().close()
            samples.append(dict(setup_ms))
    b = fresh()
    if len(samples) > 1:
        import numpy as _np
        for k in ("populate_ms", "populate_in_library_ms", "prepare_ms"):
            setup_ms[k] = float(_np.median([s_[k] for s_ in samples[1:]]))
        setup_ms["samples"] = len(samples) - 1
        setup_ms["prepare_cold_ms"] = float(_np.median([s_["prepare_ms"] for s_ in cold[1:]]))
        setup_ms["prepare_note"] = "prepare_ms = structure cache hit (same topology as the call before); prepare_cold_ms = cache emptied first"
        setup_ms["struct_cache_hits_misses"] = list(chain_bundle.struct_cache_stats())
    if bw is not None:
        bw.Compute(PREWARM)
        bw.Compute(args.warmup)
        bw.close()
    barrier()
    t0 = time.perf_counter()
    rc = b.Compute(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if rc != args.steps:
        raise SystemExit("bench: Compute ran %d of %d iterations (%s)" % (rc, args.steps, chain_bundle.last_error()))
    logs = b.IterLogs()
    tm_run = b.Timing()
    trials = sum(l["trials"] for l in logs)
    chi_first, chi_last = logs[0]["chi2_start"], logs[-1]["chi2_end"]
    b.close()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.debug_single_device else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # the adjustment MCPTAM makes NEXT (src/MapMaker.cc:225-230, 283-287): the outliers this adjustment flagged are erased
    # (MapMakerServerBase::HandleOutliers, src/MapMakerServerBase.cc:1198-1238), poses and points stay where it left them, and a fresh
    # ChainBundle adjusts again -- a topology the structure cache has NOT seen, but a subset of one it has (include/mcp_ba.h, near miss)
    after_removal = None
    if world == 1 and args.warmup > 0:
        try:
            import copy
            import numpy as _np
            b1 = chain_bundle.ChainBundle(problem.cams, True, True, False, disable_convergence=True, device=local_rank)
            ids1 = problem.populate(b1)
            b1.Compute(args.steps)
            outs = b1.GetOutlierMeasurements()
            R1, t1_ = b1.GetPoses(ids1["mkf"]); X1 = b1.GetPoints(ids1["point"])
            b1.close()
            q = synth.erase_measurements(problem, outs, ids1)
            q.base_R, q.base_t, q.pt_x = _np.array(R1), _np.array(t1_), _np.array(X1)
            near0 = chain_bundle.struct_cache_near_hits()
            calls, preps, its = [], [], []
            for _ in range(4):
                b2 = chain_bundle.ChainBundle(q.cams, True, True, False, disable_convergence=True, device=local_rank)
                q.populate(b2)
                ta = time.perf_counter(); b2.Prepare(); tb = time.perf_counter(); rc2 = b2.Compute(args.steps); tc = time.perf_counter()
                calls.append(b2.abi_seconds + (tc - ta)); preps.append((tb - ta) * 1e3); its.append((tc - tb) * 1e3)
                b2.close()
            med = float(_np.median(calls[1:]))
            after_removal = {"value": args.steps / med, "unit": "LM iterations/s over a whole call of %d iterations" % args.steps, "measurements_erased": len(outs),
                             "measurements": q.n_meas, "prepare_ms": float(_np.median(preps[1:])), "iterations_ms": float(_np.median(its[1:])), "iterations_run": rc2,
                             "structure_adopted_from_the_superset": chain_bundle.struct_cache_near_hits() - near0, "calls_timed": len(calls) - 1,
                             "note": "call 2 of {call 1; erase call 1's Tukey outliers as MapMakerServerBase::HandleOutliers does; call 2 from call 1's state}: "
                                     "bulk replay through the C ABI + Prepare() (near miss: the cached structure of the un-erased map, erased measurements weighted 0) + the iterations"}
        except Exception as exc:
            after_removal = {"error": repr(exc)}

    result = None
    if rank == 0:
        strong = args.scaling == "strong" and world > 1
        value = (1 if strong else world) * args.steps / dt
        result = {
            "metric": "ChainBundle LM iters/sec (4-cam, 200 MKF, 50k pts, 400k meas) @1/2/4/8 GPU",
            "value": value, "unit": "LM iterations/s of the one partitioned map" if strong else "LM iterations/s (x map shards)", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic (seed %d, SURVEY.md 8(d) generator)" % synth.DEFAULT_SEED,
            "config": {"workload": "%s: %d cams, %d MKF, %d points, %d measurements per rank" % (
                args.config, len(problem.cams), problem.n_mkf, problem.n_points, problem.n_meas),
                "trials_per_iteration": trials / args.steps, "trial_solves_per_s": world * trials / dt, "parallelism": ("one map partitioned over %d ranks by source MKF, poses replicated" if strong else "points sharded x%d, poses replicated") % world, "allreduce_transport": transport,
                "chi2_first": chi_first, "chi2_last": chi_last,
                "device_prewarm_iterations": PREWARM if args.warmup > 0 else 0,      # untimed, on a separate handle, before the W warm-up iterations
                "setup_outside_timed_region": dict(setup_ms, note="once per BundleAdjust call: C-ABI replay of the map (populate) and "
                                                   "host structure build + PCIe upload (prepare); not part of `value`")},
        }
        # the same K iterations as one BundleAdjust() call pays for them: bulk replay through the C ABI + Prepare() + the iterations
        call_s = (setup_ms["populate_in_library_ms"] + setup_ms["prepare_ms"]) * 1e-3 + dt
        result["value_including_setup"] = {"value": world * args.steps / call_s, "unit": "LM iterations/s over a whole call of %d iterations" % args.steps,
                                           "setup_ms": setup_ms["populate_in_library_ms"] + setup_ms["prepare_ms"], "iterations_ms": dt * 1e3}
        if "prepare_cold_ms" in setup_ms:
            cold_s = (setup_ms["populate_in_library_ms"] + setup_ms["prepare_cold_ms"]) * 1e-3 + dt
            result["value_including_setup"]["value_cold_structure"] = world * args.steps / cold_s
            result["value_including_setup"]["setup_cold_ms"] = setup_ms["populate_in_library_ms"] + setup_ms["prepare_cold_ms"]
            result["value_including_setup"]["note"] = "value: the call repeats the topology of the call before (structure cache hit); value_cold_structure: a topology this process has not seen"
        if after_removal is not None:
            result["value_after_outlier_removal"] = after_removal
        if world > 1:
            # what the LM loop put on the wire, per iteration (the first iteration's extras and the final statistics included):
            # main lane = the collectives the trial path waits for, speculative lane = beside it on the second stream
            result["config"]["collectives_per_iteration"] = {
                "main_lane": tm_run["n_collectives_main"] / args.steps, "speculative_lane": tm_run["n_collectives_spec"] / args.steps,
                "main_lane_bytes": tm_run["collective_bytes_main"] / args.steps, "speculative_lane_bytes": tm_run["collective_bytes_spec"] / args.steps,
                "one_collective_medians": tm_run["n_median_fast"], "reduced_system_solves": tm_run["n_solves"]}
        if args.debug_single_device:
            result["config"]["debug"] = "all ranks share GPU 0, gloo transport: code-path check only, not a measurement"
    # per-stage HIP-event timing of the same run shape (separate pass so the events do not perturb `value`)
    if not args.no_roofline:
        bp = fresh(profile=True)
        bp.Compute(args.steps)
        tm = bp.Timing()
        np_ = 6 * int((~problem.base_fixed).sum())
        bp.close()
        if rank == 0:
            roofs = stage_rooflines(tm, problem.n_meas, problem.n_points, np_, tm["n_linearize"], tm["n_trials"], tm["n_solves"])
            stage_ms = {k: tm[k] for k in ("eval_ms", "select_ms", "linearize_ms", "schur_ms", "cholesky_ms", "solve_ms", "update_ms")}
            dom = max(roofs.items(), key=lambda kv: kv[1]["avg_ms"] * kv[1]["launches"])
            r = dict(dom[1])
            r["kernel"] = dom[0]
            result["roofline"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_ms")}
            result["roofline"]["traffic_source"] = (TRAFFIC_FILE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of this command; bytes = (2*FETCH + WRITE)*1024 "
                                                    "per launch, summed over the launches of the stage)") if r.get("traffic") is not None else "no PMC pass of this round's kernels in " + TRAFFIC_FILE
            result["config"]["reduced_system_solves"] = tm["n_solves"]
            result["config"]["trials_served_speculatively"] = tm["n_spec_hits"]
            result["config"]["persist_fallbacks"] = tm_run["n_persist_fallbacks"] + tm["n_persist_fallbacks"]      # 0 in a healthy run (one-launch factorisation never timed out)
            result["config"]["factorisation_chains"] = tm_run.get("chol_chains", 0)      # chains of block columns the one-launch factorisation walks beside each other (arcs of the cut + the separator; 1 = the poses in add order)
            result["stages"] = {"ms_total": stage_ms, "per_stage": {k: {kk: v[kk] for kk in ("bound", "achieved", "unit", "frac", "avg_ms", "launches", "traffic", "limited_by", "mfma_executed", "plan_basis") if kk in v} for k, v in roofs.items()}}
    # CPU baselines on this box's host cores, same map, same run; the oracle's iteration log doubles as the
    # parity check of the GPU run that was just timed
    if rank == 0 and world == 1 and args.cpu_iters > 0:
        build_note = native_oracle()
        from oracle import OracleBundle

        def cpu_run(solver, threads, iters):
            o = OracleBundle(problem.cams, True, True, False)
            o.DisableConvergence(True)
            nthr = o.SetSolver(solver, threads)
            oids = problem.populate(o)
            o.Prepare()
            t0 = time.perf_counter()
            rc = o.Compute(iters)
            return o, oids, rc, time.perf_counter() - t0, nthr

        ncores = usable_cores()
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")      # idle OpenMP threads sleep instead of spinning next to the GPU driver threads
        o, oids, rc, cdt, _ = cpu_run(0, 1, args.cpu_iters)
        result["parity_at_metric"] = parity_against_oracle(problem, chain_bundle, local_rank, logs, o, oids, rc)
        # the parity above ran on the box-native build of the oracle (-O3 -march=native -fopenmp); the tests use the portable one:
        # both builds compile with -ffp-contract=off and must agree bit for bit -- checked on a small map in a child process
        try:
            import subprocess
            code = ("import sys, json; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests');"
                    "from mcptam_amd import synth; from oracle import OracleBundle; from helpers import run_bundle;"
                    "p = synth.make_config('c2', n_mkf=12, n_points=1500); r = run_bundle(OracleBundle(p.cams, True, True, False), p, 4);"
                    "print(json.dumps([r['logs'], r['t'].tolist()]))") % (ROOT, ROOT)
            env_p = dict(os.environ); env_p.pop("ORC_LIB", None)
            a_ = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env_p).stdout.strip().splitlines()[-1]
            b_ = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ)).stdout.strip().splitlines()[-1]
            result["parity_at_metric"]["oracle_builds_agree_bitwise"] = bool(a_ == b_)
        except Exception as exc:
            result["parity_at_metric"]["oracle_builds_agree_bitwise"] = "not checked: %r" % (exc,)
        variants = {"schur_1thread": {"value": rc / cdt, "unit": "LM iterations/s", "cores": 1, "kind": "port",
                                      "sample": "%d LM iterations; oracle/ba_oracle.c as the parity tests use it: points eliminated, dense Cholesky of the "
                                                "reduced system, full sort for the median" % rc}}
        _, _, rca, dta, _ = cpu_run(1, 1, args.cpu_iters)
        base_a = {"value": rca / dta, "unit": "LM iterations/s", "cores": 1, "kind": "port",
                  "sample": "%d LM iterations of the same %d-measurement map; variant A 'reference-shaped' (oracle/ba_baseline.inc): the un-marginalised "
                            "%d-unknown system factored per trial by a simplicial sparse L D L^T (points ordered first), 1 thread, full sort for the "
                            "median -- a PROXY for what the reference does with g2o + CHOLMOD (src/ChainBundle.cc:1150-1158), which cannot be built here; %s"
                            % (rca, problem.n_meas, 3 * problem.n_points + 6 * int((~problem.base_fixed).sum()), build_note),
                  "host_cores_available": ncores}
        variants["A_unmarginalised_sparse_ldlt_1thread"] = base_a
        # A2: what a supernodal CHOLMOD does with this structure on one thread -- AMD eliminates the 3x3 point blocks first (that IS the
        # Schur complement), the dense pose part goes through blocked (BLAS-3 shaped) kernels: variant B's tiled code on ONE thread
        _, _, rc2, dt2, _ = cpu_run(2, 1, args.cpu_iters)
        variants["A2_supernodal_shaped_1thread"] = {"value": rc2 / dt2, "unit": "LM iterations/s", "cores": 1, "kind": "port",
                                                    "sample": "%d LM iterations; points eliminated first + tiled dense Cholesky of the pose block, one thread, quick-select median: "
                                                              "the operation count and blocking of a supernodal sparse Cholesky (CHOLMOD) on this system" % rc2}
        if variants["A2_supernodal_shaped_1thread"]["value"] > base_a["value"]:
            base_a = dict(variants["A2_supernodal_shaped_1thread"], host_cores_available=ncores)     # the stronger single-thread figure is the denominator
        # B: the thread count that serves it best on this box (os.cpu_count() can exceed what the container may use; more
        # threads than cores only adds barrier time), a short probe per candidate, then the full sample at the best one
        tried = {}
        cands = sorted({c for c in (8, 16, 32, 64, 128, usable_cores()) if 1 <= c <= usable_cores()})
        for c in cands:
            _, _, rcp, dtp, nthr = cpu_run(2, c, 2)
            tried[nthr] = rcp / dtp
        best = max(tried, key=tried.get)
        _, _, rcb, dtb, nthr = cpu_run(2, best, args.cpu_iters)
        variants["B_schur_openmp_all_cores"] = {"value": rcb / dtb, "unit": "LM iterations/s", "cores": nthr, "kind": "port",
                                                "sample": "%d LM iterations; variant B 'best CPU': Schur, OpenMP over points / measurements with per-thread block "
                                                          "accumulators merged in thread order, tiled dense Cholesky over OpenMP, quick-select median; %d threads "
                                                          "(best of the probed thread counts)" % (rcb, nthr),
                                                "probe_it_per_s_by_threads": tried, "host_cores_usable": usable_cores()}
        result["cpu_baseline"] = base_a                 # the denominator SURVEY.md 8(d) names for the >= 10x target
        result["cpu_baseline_variants"] = variants
        result["speedup_vs_cpu"] = {k: result["value"] / v["value"] for k, v in variants.items()}
    # secondary line: the call MCPTAM makes most -- BundleAdjustRecent (src/BundleAdjusterBase.cc:188-265): the newest MKF + its 3
    # neighbours free, every other MKF that sees their points fixed, 10 iterations; timed as WHOLE calls (fresh handle, bulk replay,
    # Prepare, Compute(10), read-back of poses and points), the way BundleAdjusterMulti::BundleAdjust runs it
    if rank == 0 and world == 1 and args.config == "metric" and args.cpu_iters > 0:      # (--cpu-iters 0 = the profiling passes: headline workload only)
        try:
            import numpy as np
            w = synth.recent_window(problem)
            calls = []
            parts = {"create_ms": [], "populate_in_library_ms": [], "prepare_ms": [], "compute_ms": [], "readback_in_library_ms": [], "destroy_ms": []}
            for rep_ in range(12):
                t0 = time.perf_counter()
                bw_ = chain_bundle.ChainBundle(w.cams, True, True, False, device=local_rank)
                ids = w.populate(bw_)
                t1 = time.perf_counter()
                bw_.Prepare()
                t2 = time.perf_counter()
                rcw = bw_.Compute(10)
                t3 = time.perf_counter()
                bw_.GetPoses(ids["mkf"]); bw_.GetPoints(ids["point"]); bw_.GetOutlierMeasurements()
                t4 = time.perf_counter()
                abi_c, abi_w, abi_r = bw_.abi_create_seconds, bw_.abi_seconds, bw_.abi_read_seconds
                bw_.close()
                t5 = time.perf_counter()
                if rep_ >= 2:
                    # (the Python marshalling around the Add* / Get* entries is the harness, not the call: a native caller pays the library's time)
                    calls.append(abi_c + abi_w + (t3 - t1) + abi_r + (t5 - t4))
                    parts["create_ms"].append(abi_c * 1e3); parts["populate_in_library_ms"].append(abi_w * 1e3); parts["prepare_ms"].append((t2 - t1) * 1e3)
                    parts["compute_ms"].append((t3 - t2) * 1e3); parts["readback_in_library_ms"].append(abi_r * 1e3); parts["destroy_ms"].append((t5 - t4) * 1e3)
            med = float(np.median(calls))
            result["recent_window"] = {"workload": "BundleAdjustRecent window of the metric map: %d MKF (%d free), %d points, %d measurements, 10 LM iterations per call; a call = create, bulk Add*, Prepare, Compute(10), Get*, destroy"
                                       % (w.n_mkf, int((~w.base_fixed).sum()), w.n_points, w.n_meas),
                                       "calls_per_s": 1.0 / med, "ms_per_call": med * 1e3, "iterations_run": rcw,
                                       "ms_median": {k: float(np.median(v)) for k, v in parts.items()}, "calls_timed": len(calls)}
        except Exception as exc:
            result["recent_window"] = {"error": repr(exc)}
    # secondary lines: the other ChainBundle configurations of BASELINE.json on this one device -- c2, c4 as a whole map, and one rank's
    # eighth of c4 through the multi-rank machine (scripts/bench_secondary.py); the inputs of DESIGN.md 6's scaling model
    if rank == 0 and world == 1 and args.config == "metric" and args.cpu_iters > 0:
        try:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import bench_secondary
            result["secondary"] = {}
            for name_ in ("c2", "c4", "c4_rank_shard", "metric_forced_multi"):
                try:
                    result["secondary"][name_] = bench_secondary.run(name_, steps=(args.steps if name_ == "metric_forced_multi" else 10), device=local_rank)      # (the machine's overhead against the headline: same iteration count)
                except Exception as exc:
                    result["secondary"][name_] = {"error": repr(exc)}
        except Exception as exc:
            result["secondary"] = {"error": repr(exc)}
    # secondary line: the per-frame Tracker path (BASELINE config c3), GPU through the C ABI next to the scalar CPU port
    if rank == 0 and world == 1 and not args.no_tracker and args.cpu_iters > 0:
        try:
            sys.path.insert(0, os.path.join(ROOT, "scripts"))
            import bench_tracker
            result["tracker_c3"] = bench_tracker.main(frames=20, cpu_frames=2)
        except Exception as exc:       # the headline line must not depend on the secondary one
            result["tracker_c3"] = {"error": repr(exc)}
        try:
            result["tracker_c5"] = bench_tracker.main_c5(frames=10, cpu_frames=1)       # eight 1280x960 cameras, 8k points, on this one device
        except Exception as exc:
            result["tracker_c5"] = {"error": repr(exc)}
    sys.stdout.flush()
    if rank == 0:
        os.write(real_stdout, (json.dumps(result) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
