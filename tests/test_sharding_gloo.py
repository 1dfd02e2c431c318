"""N > 1 path on CPU: two gloo ranks each own one time window (mc_set_shard semantics, Problem::finalize), accumulate
their partial upper-triangular J^T J / -J^T e / cost with the product's term math (tests/hostprobe, g++ build of
mc_terms.h) into the GLOBAL CCS numbering, all-reduce, and compare the sum with the oracle's serial buildSystem.
This is bench.py's combination step (all_reduce of Cx, rhs, J) with the HIP kernels replaced by their host twin."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    from multical_amd import synth
    from oracle_api import OracleApi
    from probe_api import ProbeApi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = synth.SynthConfig(duration=1.0, n_cams=2, n_imus=1, n_lidars=1, lidar_points_per_frame=10, corner_stride=16,
                            huber_accel=1.0, blake_zisserman_cam=2, bias_motion_terms=False)
    spec, H, W, meas = synth.make_problem(cfg, OracleApi)
    p = ProbeApi()
    p.set_shard(rank, world)
    spec.emit(p)
    p.finalize()
    Cp, Ci = p.pattern()
    n = p.j_cols
    Hd = np.zeros((n, n))
    rhs = np.zeros(n)
    J = 0.0
    mine = p.my_terms()
    for t in np.nonzero(mine)[0]:
        Jt, e, chi2 = p.term(int(t), True)
        Hd += Jt.T @ Jt
        rhs -= Jt.T @ e
    # values in the global CCS order (upper triangle)
    cols = np.repeat(np.arange(n), np.diff(Cp))
    Cx_local = Hd[Ci, cols].copy()
    # bench.py's combination: only the entries more than one rank can write go through the all-reduce
    sidx = p.shared_indices().astype(np.int64)
    buf = torch.from_numpy(Cx_local[sidx].copy())
    dist.all_reduce(buf)
    Cx_local[sidx] = buf.numpy()
    # everything else must be owned by exactly one rank: count the ranks with a non-zero there
    private = np.ones(Cx_local.size, bool)
    private[sidx] = False
    touched = torch.from_numpy(((Cx_local != 0.0) & private).astype(np.float64))
    dist.all_reduce(touched)
    assert float(touched.max()) <= 1.0, "an entry outside the shared set is written by two ranks"
    # assemble the whole matrix for the comparison: shared part from any rank, private part = sum (one owner each)
    priv = torch.from_numpy(np.where(private, Cx_local, 0.0))
    dist.all_reduce(priv)
    Cx = torch.from_numpy(np.where(private, priv.numpy(), Cx_local))
    rhs_t = torch.from_numpy(rhs)
    cnt = torch.tensor([float(mine.sum())], dtype=torch.float64)
    dist.all_reduce(rhs_t)
    dist.all_reduce(cnt)
    if rank == 0:
        np.savez(out_path, Cx=Cx.numpy(), rhs=rhs_t.numpy(), count=cnt.numpy(), n_shared=sidx.size, nnz=Cx_local.size, kinds=np.array([p.term_kind(t) for t in range(p.n_terms)]),
                 n_terms=p.n_terms)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_partial_systems_sum_to_the_oracle(tmp_path):
    from multical_amd import synth
    from oracle_api import OracleApi
    out = str(tmp_path / "sum.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    assert int(got["count"][0]) == int(got["n_terms"])          # every term owned by exactly one rank
    assert 0 < int(got["n_shared"]) < 0.5 * int(got["nnz"])     # the all-reduce moves a fraction of the matrix
    cfg = synth.SynthConfig(duration=1.0, n_cams=2, n_imus=1, n_lidars=1, lidar_points_per_frame=10, corner_stride=16,
                            huber_accel=1.0, blake_zisserman_cam=2, bias_motion_terms=False)
    spec, H, W, meas = synth.make_problem(cfg, OracleApi)
    o = OracleApi()
    spec.emit(o)
    o.finalize()
    o.evaluate_error(True)
    Cp, Ci, Cx, rhs = o.build_system(True)
    want_Cx, want_rhs = Cx, rhs
    scale = np.abs(want_Cx).max()
    assert np.abs(got["Cx"] - want_Cx).max() <= 1e-9 * scale
    assert np.abs(got["rhs"] - want_rhs).max() <= 1e-9 * np.abs(want_rhs).max()


# ---------------------------------------------------------------------------------------------------------------------
# The N-rank ITERATION (mc_optimize_sharded; Optimizer2.cpp:183-275 with applyStateUpdate / revert on every rank, :292-320)
# on CPU ranks: the product's minimiser loop (csrc/host/lm_core.hpp) and factorisation over the host twin of the kernels.
def _lm_cfg():
    from multical_amd import synth
    # well excited (motion_speed) and run to convergence: the converged estimate does not remember the summation order
    return synth.SynthConfig(duration=3.0, n_cams=2, n_imus=1, n_lidars=1, lidar_points_per_frame=20, corner_stride=8,
                             huber_accel=1.0, blake_zisserman_cam=2, motion_speed=8.0)


LM_ARGS = dict(max_iterations=30, lambda0=10.0, dx_tol=1e-5, ratio_tol=1e-6)   # calibrator.py:27-58's thresholds


def _state_of(api, spec, ids, H, W):
    out = []
    for h, v in sorted(ids.items()):
        if isinstance(v, int) and spec.ops[h][0] == "dv":
            out.append(api.dv_params(v))
    first = api.spline_first_dv(ids[H["pose"]][1])
    for i in range(W["pose_coeffs"].shape[1]):
        out.append(api.dv_params(first + i))
    return np.concatenate(out)


def _lm_worker(rank, world, port, out_path):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    from multical_amd import synth
    from oracle_api import OracleApi
    from probe_api import ProbeApi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec, H, W, meas = synth.make_problem(_lm_cfg(), OracleApi)
    p = ProbeApi()
    p.set_shard(rank, world)
    ids = spec.emit(p)
    p.finalize()
    calls = dict(allreduce=0, allreduce_doubles=0, broadcast=0)

    def allreduce(a):
        t = torch.from_numpy(a)
        dist.all_reduce(t)
        calls["allreduce"] += 1
        calls["allreduce_doubles"] += a.size

    def broadcast(a, root):
        dist.broadcast(torch.from_numpy(a), src=root)
        calls["broadcast"] += 1
    r = p.optimize(allreduce=allreduce, broadcast=broadcast, **LM_ARGS)
    x = _state_of(p, spec, ids, H, W)
    # every rank holds the same replica of the state, bit for bit
    t = torch.from_numpy(x.copy())
    dist.broadcast(t, src=0)
    assert np.array_equal(t.numpy(), x), "the ranks' replicas of the state have parted"
    if rank == 0:
        np.savez(out_path, x=x, res=np.array([r["JStart"], r["JFinal"], r["iterations"], r["failedIterations"], r["linearSolverFailure"]]),
                 calls=np.array([calls["allreduce"], calls["allreduce_doubles"], calls["broadcast"]]), nnz=p.nnz, n=p.j_cols)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_levenberg_marquardt_matches_the_one_rank_run(tmp_path):
    """Two gloo ranks, each with one time window, run the N-rank minimiser: same accepted / rejected sequence, same cost and
    estimates within 1e-7 as the unsharded run of the same loop -- and as the oracle's restatement of Optimizer2 + LM.  The
    value array crosses the ranks only for ADOPTED linearisations (one all-reduce each), the cost once per trial."""
    from multical_amd import synth
    from oracle_api import OracleApi
    from oracle_lm import optimize_lm
    from probe_api import ProbeApi
    out = str(tmp_path / "lm.npz")
    mp.spawn(_lm_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    spec, H, W, meas = synth.make_problem(_lm_cfg(), OracleApi)
    one = ProbeApi()
    ids = spec.emit(one)
    one.finalize()
    r1 = one.optimize(**LM_ARGS)
    x1 = _state_of(one, spec, ids, H, W)
    res = got["res"]
    assert (int(res[2]), int(res[3]), int(res[4])) == (r1["iterations"], r1["failedIterations"], r1["linearSolverFailure"]), (res, r1)
    assert r1["failedIterations"] >= 0 and r1["iterations"] >= 3
    # the two runs sum the same terms in different orders; over a dozen iterations the last bits grow to ~1e-8 of the cost
    assert abs(res[0] - r1["JStart"]) <= 1e-9 * r1["JStart"] and abs(res[1] - r1["JFinal"]) <= 1e-7 * r1["JFinal"]
    assert np.abs(got["x"] - x1).max() <= 1e-7, np.abs(got["x"] - x1).max()
    assert r1["JFinal"] < 0.2 * r1["JStart"]
    # collectives: one 2-double all-reduce per trial (+ the initial one), two per adopted linearisation (values, gradient)
    n_ar, n_doubles, n_bc = (int(v) for v in got["calls"])
    trials = r1["iterations"] + 1
    adopted = (n_ar - trials) // 2
    assert n_ar == trials + 2 * adopted and 1 <= adopted <= trials
    assert n_doubles == 2 * trials + adopted * (int(got["nnz"]) + int(got["n"]))
    assert n_bc >= r1["iterations"]
    # and the oracle's Optimizer2 + LevenbergMarquardtTrustRegionPolicy on the whole problem
    o = OracleApi()
    oids = spec.emit(o)
    o.finalize()
    ro = optimize_lm(o, LM_ARGS["max_iterations"], LM_ARGS["lambda0"], LM_ARGS["dx_tol"], LM_ARGS["ratio_tol"])
    assert (ro["iterations"], ro["failedIterations"]) == (r1["iterations"], r1["failedIterations"]), (ro, r1)
    assert abs(ro["JFinal"] - res[1]) <= 1e-7 * ro["JFinal"]
    xo = _state_of(o, spec, oids, H, W)
    assert np.abs(got["x"] - xo).max() <= 1e-7, np.abs(got["x"] - xo).max()
