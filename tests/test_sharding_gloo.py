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
