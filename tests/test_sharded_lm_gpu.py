"""mc_optimize_sharded on the GPU, rehearsed on ONE card: two processes, each with one time-window shard of the problem on
cuda:0, the collectives through gloo and host memory (RCCL needs a second device; bench.py's MC_BENCH_REHEARSE path does the
same).  Same accepted / rejected sequence, cost and estimates (1e-7) as mc_optimize on the unsharded context -- the N-rank
form of Optimizer2::optimize (aslam_backend/src/Optimizer2.cpp:183-275, 292-320)."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_sharding_gloo import LM_ARGS, _free_port, _lm_cfg, _state_of

pytestmark = pytest.mark.gpu

KW = dict(maxIterations=LM_ARGS["max_iterations"], levenbergMarquardtLambdaInit=LM_ARGS["lambda0"],
          convergenceDeltaX=LM_ARGS["dx_tol"], convergenceJDescentRatioThreshold=LM_ARGS["ratio_tol"])


def _worker(rank, world, port, out_path):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MC_SERIAL_LAUNCH"] = "1"
    from multical_amd import synth
    from multical_amd.backend import HipApi, torch_collectives
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    torch.cuda.set_device(0)
    spec, H, W, meas = synth.make_problem(_lm_cfg(), lambda: HipApi(0))
    g = HipApi(0)
    g.set_shard(rank, world)
    ids = spec.emit(g)
    g.finalize()
    allreduce, broadcast = torch_collectives(torch.device("cuda", 0), through_host=True)
    r = g.optimize_sharded(allreduce, broadcast, **KW)
    x = _state_of(g, spec, ids, H, W)
    t = torch.from_numpy(x.copy())
    dist.broadcast(t, src=0)
    assert np.array_equal(t.numpy(), x), "the ranks' replicas of the state have parted"
    if rank == 0:
        np.savez(out_path, x=x, res=np.array([r["JStart"], r["JFinal"], r["iterations"], r["failedIterations"], r["linearSolverFailure"]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_minimiser_on_two_ranks_matches_the_unsharded_one(tmp_path):
    from multical_amd import synth
    from multical_amd.backend import BackendError, HipApi
    out = str(tmp_path / "lm_gpu.npz")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    spec, H, W, meas = synth.make_problem(_lm_cfg(), lambda: HipApi(0))
    g = HipApi(0)
    ids = spec.emit(g)
    r1 = g.optimize(**KW)
    x1 = _state_of(g, spec, ids, H, W)
    res = got["res"]
    assert (int(res[2]), int(res[3]), int(res[4])) == (r1["iterations"], r1["failedIterations"], r1["linearSolverFailure"]), (res, r1)
    assert r1["iterations"] >= 3 and r1["JFinal"] < 0.2 * r1["JStart"]
    assert abs(res[0] - r1["JStart"]) <= 1e-9 * r1["JStart"] and abs(res[1] - r1["JFinal"]) <= 1e-7 * r1["JFinal"]
    assert np.abs(got["x"] - x1).max() <= 1e-7, np.abs(got["x"] - x1).max()
    # mc_optimize itself refuses a sharded context (it would factorise a partial system)
    s = HipApi(0)
    s.set_shard(0, 2)
    spec.emit(s)
    s.finalize()
    with pytest.raises(BackendError, match="mc_optimize_sharded"):
        s.optimize(**KW)
