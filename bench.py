#!/usr/bin/env python
"""bench.py -- error-terms/s of one Gauss-Newton iteration's evaluateError + buildSystem on MI355X.

A step = LinearSystemSolver::evaluateError followed by BlockCholeskyLinearSystemSolver::buildSystem over the whole
synthetic problem (inputs resident in HBM), plus -- for N > 1 -- the RCCL all-reduce of the partial H / b / J.
Workload: BASELINE.json configs[3], the full Multical problem (3 IMUs + 4 cameras + 2 LiDARs, 6 boards, time
offsets, 120 s) that north_star quotes its target on; with N GPUs the sequence is N x 120 s long and each rank owns one
time window ("weak" scaling).  The host sparse Cholesky is not part of the step (north_star: timed separately; see
--solve).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector = fp64 matrix peak (AMD spec sheet; MI355X_MICROARCH.md lists no fp64 row)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=3, help="index into BASELINE.json configs (1, 3 or 4)")
    ap.add_argument("--duration", type=float, default=None, help="override the sequence length per GPU [s]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-seconds", type=float, default=10.0)
    ap.add_argument("--no-solve", dest="solve", action="store_false", help="skip the host sparse Cholesky solve (N = 1: timed after the steps, reported apart in config.host_cholesky_*)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from multical_amd import synth
    from multical_amd.backend import HipApi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # Rehearsal aid for a one-GPU box: MC_BENCH_REHEARSE=1 puts every rank on cuda:0 and exchanges through gloo, so the
    # N > 1 code path (shards, exchange buffer, timing) can be run where RCCL has no second device to talk to.
    rehearse = world > 1 and os.environ.get("MC_BENCH_REHEARSE") == "1"
    if world > 1:
        # N > 1: every context of this process (the unsharded one that synthesises the measurements included) launches on
        # the caller's stream; the side streams are worth 2 % and share the process's 4 hardware queues with RCCL's
        os.environ.setdefault("MC_SERIAL_LAUNCH", "1")
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    # N > 1 must fail fast, never stall: every collective carries a time limit, and a watchdog (re-armed at every phase
    # note) dumps the Python stack of every thread and exits non-zero when a phase makes no progress -- torchrun then
    # takes the other ranks down.  Limits: MC_BENCH_PHASE_TIMEOUT seconds per phase (default 180).
    import datetime
    import faulthandler
    phase_timeout = float(os.environ.get("MC_BENCH_PHASE_TIMEOUT", "180"))
    if world > 1:
        pg_timeout = datetime.timedelta(seconds=phase_timeout)
        if rehearse:
            dist.init_process_group("gloo", timeout=pg_timeout)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=pg_timeout)

    verbose = os.environ.get("MC_BENCH_VERBOSE") == "1" or world > 1
    trace = os.environ.get("MC_BENCH_TRACE") == "1"      # per-call notes with a device synchronise in between (diagnosis)

    def note(msg):
        if world > 1:
            faulthandler.cancel_dump_traceback_later()
            faulthandler.dump_traceback_later(phase_timeout, exit=True)
        if verbose:
            print("[bench rank %d %s] %s" % (rank, time.strftime("%H:%M:%S"), msg), file=sys.stderr, flush=True)

    def tr(msg):
        if trace:
            torch.cuda.synchronize()
            note(msg)

    base = synth.baseline_config(args.config)
    per_gpu = args.duration if args.duration is not None else base.duration
    cfg = synth.baseline_config(args.config, duration=per_gpu * world)
    t0 = time.time()
    W = synth.make_world(cfg)
    def synth_api():
        # forward model at the truth, on this GPU.  With shards every rank synthesises the measurements of ITS OWN time window only
        # (same cuts: they depend on the terms' times and counts, not on the state); the others' stay zero and are never read
        a = HipApi(local_rank)
        if world > 1:
            a.set_shard(rank, world)
        return a
    meas = synth.synthesize(W, synth_api, noise=True)
    spec, H = synth.build_spec(W, synth.initial_state(W), meas)
    api = HipApi(local_rank)
    if world > 1:
        api.set_shard(rank, world)
    spec.emit(api)
    api.finalize()
    if world > 1:
        # dCx / dE below start out zero and only mc_unpack_shared touches them between passes: a pass may clear just the
        # values this rank writes (the whole global Cx is 157 MB x world)
        api.set_clear_mode(True)
    t_setup = time.time() - t0
    note("set-up done in %.1f s, %d terms, nnz %d" % (t_setup, api.n_terms, api.nnz))
    ranks_seen = None
    if world > 1:
        mine = dict(rank=rank, device=torch.cuda.current_device(), name=torch.cuda.get_device_name(), pid=os.getpid())
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, mine)
        assert dist.get_world_size() == world and sorted(r["rank"] for r in ranks_seen) == list(range(world)), ranks_seen
        note("all %d ranks present" % world)

    dev = torch.device("cuda", local_rank)
    dCx = torch.zeros(api.nnz, dtype=torch.float64, device=dev)
    dE = torch.zeros(api.j_rows, dtype=torch.float64, device=dev)     # the stacked weighted residual evaluateError returns
    stream = torch.cuda.current_stream().cuda_stream
    # One exchange buffer per step: [ Cx entries shared between ranks | rhs entries shared between ranks | J ].  J is written
    # straight into it by the entry points, the shared entries are gathered by mc_pack_shared / mc_pack_shared_rhs: ONE
    # all-reduce per Gauss-Newton iteration, of the entries more than one rank writes and nothing else.
    n_shared = len(api.shared_indices()) if world > 1 else 0
    n_shared_rhs = len(api.shared_rhs_indices()) if world > 1 else 0
    xbuf = torch.zeros(n_shared + n_shared_rhs + 1, dtype=torch.float64, device=dev)
    dshared, dshared_rhs, dJ = xbuf[:n_shared], xbuf[n_shared:n_shared + n_shared_rhs], xbuf[n_shared + n_shared_rhs:]
    drhs = torch.zeros(api.j_cols, dtype=torch.float64, device=dev)

    def exchange():
        # partial Hessians: only the entries two ranks can write (window halos, global x global blocks) are summed;
        # every other entry already sits complete on the rank that owns its time window
        api.pack_shared(dCx.data_ptr(), dshared.data_ptr(), stream)
        api.pack_shared_rhs(drhs.data_ptr(), dshared_rhs.data_ptr(), stream)
        tr("packed")
        if rehearse and os.environ.get("MC_BENCH_GLOO_DEVICE") != "1":
            h = xbuf.cpu()            # gloo on host memory (DESIGN.md section 6 has what its device-tensor path does)
            dist.all_reduce(h)
            xbuf.copy_(h)
        else:
            dist.all_reduce(xbuf)
        tr("all-reduced")
        api.unpack_shared(dshared.data_ptr(), dCx.data_ptr(), stream)
        api.unpack_shared_rhs(dshared_rhs.data_ptr(), drhs.data_ptr(), stream)
        tr("unpacked")

    def step():
        # One Gauss-Newton / Levenberg-Marquardt iteration's hot path as this repository's minimiser (csrc/host/lm_driver.hpp)
        # runs it: ONE fused pass that returns what LinearSystemSolver::evaluateError and BlockCholeskyLinearSystemSolver::
        # buildSystem return together -- the cost J, and H (fillCCS values) and rhs at that state.
        # `e` (j_rows doubles, what LinearSystemSolver::evaluateError leaves in _e) is written too, although the block solver
        # never reads it: the reference's evaluateError does that work, so the timed step does
        api.evaluate_error_and_build_system_device(True, dJ.data_ptr(), dE.data_ptr(), dCx.data_ptr(), drhs.data_ptr(), stream)
        tr("evaluated + built")
        if world > 1:
            exchange()

    def step_without_e():
        api.evaluate_error_and_build_system_device(True, dJ.data_ptr(), None, dCx.data_ptr(), drhs.data_ptr(), stream)
        if world > 1:
            exchange()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            if rehearse:
                dist.monitored_barrier(timeout=datetime.timedelta(seconds=phase_timeout))
            else:
                dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    note("warm-up launched")
    fence()
    note("warm-up done")
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t1
    note("timed steps done")
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    api.check_status()
    # the same iteration as the two calls the reference's LinearSystemSolver interface makes (evaluateError, then buildSystem):
    # reported beside the headline
    def two_call_step():
        api.evaluate_error_device(True, dJ.data_ptr(), dE.data_ptr(), stream)
        api.build_system_device(True, dCx.data_ptr(), drhs.data_ptr(), stream)
        if world > 1:
            exchange()
    two_call_step()
    fence()
    t2 = time.perf_counter()
    for _ in range(args.steps):
        two_call_step()
    fence()
    dt_two = time.perf_counter() - t2
    if world > 1:
        tt = torch.tensor([dt_two], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_two = tt.item()
    api.check_status()
    note("two-call steps done")
    step_without_e()
    fence()
    t2 = time.perf_counter()
    for _ in range(args.steps):
        step_without_e()
    fence()
    dt_noe = time.perf_counter() - t2
    if world > 1:
        tt = torch.tensor([dt_noe], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt_noe = tt.item()
    ms_per_step = dt / args.steps * 1e3
    n_terms = api.n_terms
    value = n_terms / (dt / args.steps)

    # ---- per-kernel timing (HIP events on the launch stream, inside the C ABI) and roofline of the dominant kernel
    info = api.launch_info()
    reps = max(3, min(10, args.steps))
    ms_build = np.zeros(len(info))
    ms_eval = np.zeros(len(info))
    # the two pass types back to back, not interleaved: the rocprofv3 averages in profiles/ are over consecutive passes of one
    # type too (interleaved, every launch starts on the other pass's cache contents and measures ~7 % longer)
    api.build_system_timed(True, True)
    for _ in range(reps):
        ms_build += api.build_system_timed(True, True)
    api.build_system_timed(True, False)
    for _ in range(reps):
        ms_eval += api.build_system_timed(True, False)
    ms_build /= reps
    ms_eval /= reps
    per_kernel = {}
    for (name, nbatch, nt, nwg, fd, fl, rb), mb, me in zip(info, ms_build, ms_eval):
        if nwg == 0:
            continue
        per_kernel[name] = dict(ms=float(mb), eval_ms=float(me), launches=1, flops_dense=fd, flops_live=fl, rec_bytes=rb,
                                terms=nt, workgroups=nwg)
    # buildSystem alone, as launched in the step (term families on their own streams)
    fence()
    t3 = time.perf_counter()
    for _ in range(args.steps):
        api.build_system_device(True, dCx.data_ptr(), drhs.data_ptr(), stream)
    fence()
    build_wall_ms = (time.perf_counter() - t3) / args.steps * 1e3
    note("per-kernel timing done")
    if world > 1:
        faulthandler.cancel_dump_traceback_later()     # what follows is rank-local
    # SURVEY 8(d) algorithmic bytes of one buildSystem: term records + the Hessian / gradient entries it must produce + the
    # state, read once.  A kernel's share of the Hessian / state bytes is taken in proportion to its records.
    state_bytes = 8.0 * sum(d["params"].size if k == "dv" else d["coeffs"].size for k, d in spec.ops if k in ("dv", "spline"))
    rec_total = sum(v["rec_bytes"] for v in per_kernel.values())
    out_bytes = 8.0 * (api.nnz + api.j_cols) + state_bytes
    alg_total = rec_total + out_bytes
    for v in per_kernel.values():
        v["alg_bytes"] = v["rec_bytes"] + out_bytes * v["rec_bytes"] / max(1.0, rec_total)
        v["frac"] = v["alg_bytes"] / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    # The kernel the roofline is quoted for: the LONGEST single launch of the build pass; launches within 2 % of the longest
    # count as tied (HIP-event averages move by that much from run to run) and the tie goes to the one furthest below the
    # roofline, so the choice does not flip between runs and never flatters the line.
    longest = max(v["ms"] for v in per_kernel.values())
    dom = min((k for k, v in per_kernel.items() if v["ms"] >= 0.98 * longest), key=lambda k: per_kernel[k]["frac"])
    pk = per_kernel[dom]
    avg_launch_ms = pk["ms"] / max(1, pk["launches"])
    alg_bytes = pk["alg_bytes"]
    # all launches of the pass together: algorithmic bytes of the whole build over the sum of the launch durations
    serial_ms = sum(v["ms"] for v in per_kernel.values())
    time_weighted_frac = alg_total / (serial_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    tf_dense = pk["flops_dense"] / (pk["ms"] * 1e-3) / 1e12
    tf_live = pk["flops_live"] / (pk["ms"] * 1e-3) / 1e12
    gbs = alg_bytes / (pk["ms"] * 1e-3) / 1e9
    # HBM traffic of the dominant kernel: from the committed rocprofv3 PMC passes of this same command (separate runs,
    # FETCH_SIZE x2 corrected; scripts/profile_bench.sh + scripts/summarize_profile.py) -- counters cannot be read in-process
    traffic, traffic_src = None, None
    tp = os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")
    if world == 1 and args.config == 3 and args.duration is None and os.path.exists(tp):
        with open(tp) as f:
            tj = json.load(f)
        if dom in tj.get("kernels", {}):
            traffic = tj["kernels"][dom]["fetch"] + tj["kernels"][dom]["write"]
            traffic_src = "profiles/%s_pmc_hbm_traffic.csv" % tj["tag"]
    # the same figures for every launch of the pass (the three large ones are within a few per cent of one another, so which
    # of them is "dominant" can change from run to run)
    per_kernel_roofline = {}
    for name, v in per_kernel.items():
        ab = v["rec_bytes"] + out_bytes * v["rec_bytes"] / max(1.0, rec_total)
        tr = None
        if traffic_src is not None and name in tj.get("kernels", {}):
            tr = tj["kernels"][name]["fetch"] + tj["kernels"][name]["write"]
        per_kernel_roofline[name] = dict(avg_launch_ms=round(v["ms"], 4), algorithmic_bytes=round(ab),
                                         achieved=round(ab / (v["ms"] * 1e-3) / 1e9, 2),
                                         frac=round(ab / (v["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), traffic=tr)
    # The kernels execute far fewer flops than the reference's per-term dense J^T J (shared chain matrix, 7 x 7 Gram per
    # observation, structural zeros skipped), so their floor is the HBM roofline; the fp64 rates of the REFERENCE's flop
    # count are reported beside it for SURVEY 8(d) and can exceed the fp64 peak.
    roofline = dict(bound="hbm", kernel=dom, achieved=round(gbs, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(gbs / HBM_PEAK_GBS, 5), time_weighted_frac=round(time_weighted_frac, 5),
                    kernel_choice="longest launch of the build pass; launches within 2 % tie, the tie goes to the lowest frac",
                    traffic=traffic, traffic_source=traffic_src,
                    algorithmic_bytes=round(alg_bytes), avg_launch_ms=round(avg_launch_ms, 4), launches_per_step=pk["launches"],
                    build_system=dict(algorithmic_bytes=round(alg_total), wall_ms=round(build_wall_ms, 4),
                                      achieved=round(alg_total / (build_wall_ms * 1e-3) / 1e9, 2),
                                      frac=round(alg_total / (build_wall_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)),
                    fp64=dict(peak_tflops=FP64_PEAK_TFLOPS, reference_dense_tflops=round(tf_dense, 3), reference_live_tflops=round(tf_live, 3),
                              note="flops of the reference algorithm (dense per-term J^T J incl. padded zero blocks / live blocks only) "
                                   "divided by this kernel's time; the kernel itself executes a small fraction of them"),
                    per_kernel=per_kernel_roofline,
                    kernels_ms={k: round(v["ms"], 4) for k, v in per_kernel.items()},
                    eval_kernels_ms={k: round(v["eval_ms"], 4) for k, v in per_kernel.items()})

    out = dict(metric="error-terms/sec (evaluateError + buildSystem per GN iteration)", value=value, unit="error-terms/s",
               n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True,
               scaling="weak", vs_baseline=None, dtype="f64", data="synthetic",
               config=dict(workload="BASELINE configs[%d]: %d IMUs + %d cameras + %d LiDARs, %d boards, time offsets, %.0f s per GPU"
                           % (args.config, cfg.n_imus, cfg.n_cams, cfg.n_lidars, cfg.n_boards, per_gpu),
                           error_terms=n_terms, j_rows=api.j_rows, j_cols=api.j_cols, nnz_upper=api.nnz, n_blocks=api.n_blocks,
                           build_ms=round(build_wall_ms, 4), build_kernels_serial_ms=round(float(ms_build.sum()), 4), evaluate_ms=round(float(ms_eval.sum()), 4),
                           step="one fused pass: J + e + H + rhs (mc_evaluate_error_and_build_system_device with d_e given: the stacked "
                                "residual evaluateError leaves in _e is written although the block solver never reads it)",
                           step_without_e_ms=round(dt_noe / args.steps * 1e3, 4),
                           two_call_evaluate_then_build_ms=round(dt_two / args.steps * 1e3, 4),
                           setup_s=round(t_setup, 2), shared_values=n_shared, shared_rhs_values=n_shared_rhs, ranks_seen=ranks_seen,
                           sharding="time windows, global pattern; one RCCL all-reduce of [shared Cx entries | shared rhs entries | J] per step; a pass clears only the values its rank writes (mc_set_clear_mode)" if world > 1 else "none"),
               roofline=roofline)

    if args.solve and rank == 0 and world == 1:
        # solveSystem on the host (csrc/host/block_cholesky.hpp in CHOLMOD's place): symbolic analysis once per
        # optimize(), numeric factorisation + solve per iteration -- timed apart from the step, as north_star says
        from multical_amd.backend import HipBlockCholeskyLinearSystemSolver, host_threads
        s = HipBlockCholeskyLinearSystemSolver(api)
        ts = time.perf_counter()
        s.initMatrixStructure(True)
        t_analyze = time.perf_counter() - ts
        s.buildSystem()
        s.setConstantConditioner(10.0)
        ts = time.perf_counter()
        ok, dx = s.solveSystem()          # first call: symbolic phase on the numerical pattern + factor + solve
        t_first = time.perf_counter() - ts
        t_steady = []
        for _ in range(3):                # what every further iteration pays (values change, structure does not)
            ts = time.perf_counter()
            ok, dx = s.solveSystem()
            t_steady.append(time.perf_counter() - ts)
        out["config"]["host_cholesky_s"] = round(min(t_steady), 4)
        out["config"]["host_cholesky_first_call_s"] = round(t_first, 3)
        out["config"]["host_cholesky_threads"] = host_threads()
        out["config"]["host_cholesky_analyze_s"] = round(t_analyze, 3)
        out["config"]["host_cholesky_ok"] = bool(ok)
        out["config"]["host_cholesky"] = s._host.stats()
        # the same solve through the factorisation's GPU stage (mc_schur.hip: interiors of the pose-spline band eliminated on
        # the device from the device system, reduced system on the host) -- what mc_optimize runs by default; outside `value`
        from multical_amd.backend import DeviceSchurSolver
        st = torch.cuda.current_stream().cuda_stream
        api.evaluate_error(True)
        api.build_system_device(True, dCx.data_ptr(), drhs.data_ptr(), st)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        gs = DeviceSchurSolver(api, dCx.data_ptr())
        t_plan = time.perf_counter() - ts
        if gs.planned:
            t_dev = []
            for _ in range(4):
                torch.cuda.synchronize()
                ts = time.perf_counter()
                rc, x_dev = gs.solve(dCx.data_ptr(), drhs.data_ptr(), 100.0, st)
                t_dev.append(time.perf_counter() - ts)
            info = gs.info()
            out["config"]["solve_s"] = round(min(t_dev[1:]), 4)
            out["config"]["solve"] = dict(stage="interiors on the GPU, reduced system on the host", ok=bool(rc == 0), plan_s=round(t_plan, 3),
                                          first_call_s=round(t_dev[0], 3), interiors=info["interiors"], reduced_columns=info["reduced_columns"],
                                          reduced_nnz=info["reduced_nnz"], phases_ms={k: round(v * 1e3, 3) for k, v in info["seconds"].items()})
        else:
            out["config"]["solve"] = dict(stage="host only: " + gs.why)

    if rank == 0 and not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"], out["cpu_optimised"] = cpu_baseline(args, synth, spec)
        out["speedup_vs_cpu"] = dict(reference_shaped=round(value / out["cpu_baseline"]["value"], 1),
                                     optimised=round(value / out["cpu_optimised"]["value"], 1))
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        faulthandler.cancel_dump_traceback_later()
        dist.destroy_process_group()


def cpu_baseline(args, synth, full_spec):
    """Two CPU baselines on the GPU box's host cores, both restatements -- the reference binary cannot be built here
    (SURVEY.md 8c).  The reference-shaped one runs on a bounded sample (the first `cpu_sample_seconds` of the same workload:
    its serial build would take ~90 s on the whole); the optimised one runs the WHOLE workload (~0.15 s per pass):

    reference-shaped ("port")  the CPU oracle: per-term expression-DAG walk with heap matrices, std::map JacobianContainer /
                  SparseBlockMatrix, buildSystem SERIAL on one core (BlockCholeskyLinearSystemSolver.cpp:58-72 ignores
                  nThreads), evaluateError on nproc - 1 threads like calibrator.py:33 sets it
    optimised     this repository's own formulation compiled for the host (closed forms, chunk tiles, 7 x 7 Gram + chain
                  matrix, slot tables) with OpenMP on all cores: tests/hostprobe/cpu_optimised.inc
    """
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_api import OracleApi
    from probe_api import ProbeApi
    from multical_amd.backend import HipApi
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or os.cpu_count() or 1
    cfg = synth.baseline_config(args.config, duration=args.cpu_sample_seconds)
    W = synth.make_world(cfg)
    meas = synth.synthesize(W, lambda: HipApi(0), noise=True)
    spec, _ = synth.build_spec(W, synth.initial_state(W), meas)
    o = OracleApi()
    spec.emit(o)
    o.finalize()
    nt = max(1, cores - 1)
    o.evaluate_error(True, False, nt)
    t = time.perf_counter()
    o.evaluate_error(True, False, nt)
    t_eval = time.perf_counter() - t
    t = time.perf_counter()
    o.build_only(True)
    t_build = time.perf_counter() - t
    base = dict(value=o.n_terms / (t_eval + t_build), unit="error-terms/s", cores=nt, kind="port",
                sample="first %.1f s of the same workload (%d error terms): buildSystem serial on 1 core as in the reference "
                       "(%.3f s), evaluateError on %d threads (%.3f s)" % (args.cpu_sample_seconds, o.n_terms, t_build, nt, t_eval),
                build_terms_per_s=o.n_terms / t_build, evaluate_threads=nt)
    del o
    p = ProbeApi()
    full_spec.emit(p)
    p.finalize()
    p.cpu_optimised(True, True, cores)
    reps = 5
    t = time.perf_counter()
    for _ in range(reps):
        p.cpu_optimised(True, True, cores)
    t_opt = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    p.cpu_optimised(True, True, 1)
    t_opt1 = time.perf_counter() - t
    opt = dict(value=p.n_terms / t_opt, unit="error-terms/s", cores=cores, kind="port (this repository's formulation, OpenMP)",
               sample="the whole workload (%d error terms); one fused evaluateError + buildSystem pass in %.4f s on %d threads "
                      "(%.4f s on 1 thread)" % (p.n_terms, t_opt, cores, t_opt1), one_thread_value=p.n_terms / t_opt1)
    return base, opt


if __name__ == "__main__":
    main()
