#!/usr/bin/env python
"""Rehearsal of ONE rank of `bench.py --gpus N` on a single GPU: builds the N x 120 s problem, takes shard `rank` of N,
and times set-up and the step without the collectives (there is no second GPU to talk to).  Checks that the set-up time
and the host / device memory of the N = 8 case are what the driver's 8-GPU run can afford, and that a shard's step time
equals the single-GPU step (weak scaling by construction).

    python scripts/shard_rehearsal.py --world 8 --rank 3
"""
import argparse
import json
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--rank", type=int, default=0)
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--steps", type=int, default=10)
    args = ap.parse_args()
    import torch
    from multical_amd import synth
    from multical_amd.backend import HipApi

    base = synth.baseline_config(args.config)
    cfg = synth.baseline_config(args.config, duration=base.duration * args.world)
    t0 = time.time()
    W = synth.make_world(cfg)
    meas = synth.synthesize(W, lambda: HipApi(0), noise=True)
    t_synth = time.time() - t0
    spec, H = synth.build_spec(W, synth.initial_state(W), meas)
    t_spec = time.time() - t0
    api = HipApi(0)
    api.set_shard(args.rank, args.world)
    spec.emit(api)
    t_emit = time.time() - t0
    api.finalize()
    t_setup = time.time() - t0
    dev = torch.device("cuda", 0)
    dCx = torch.zeros(api.nnz, dtype=torch.float64, device=dev)
    drhs = torch.zeros(api.j_cols, dtype=torch.float64, device=dev)
    dJ = torch.zeros(1, dtype=torch.float64, device=dev)
    n_shared = len(api.shared_indices())
    dshared = torch.zeros(max(1, n_shared), dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    dE = torch.zeros(api.j_rows, dtype=torch.float64, device=dev)
    if os.environ.get("MC_REHEARSE_FULL_CLEAR") != "1":
        api.set_clear_mode(True)

    def step():     # bench.py's step without the all-reduce
        api.evaluate_error_and_build_system_device(True, dJ.data_ptr(), dE.data_ptr(), dCx.data_ptr(), drhs.data_ptr(), stream)
        api.pack_shared(dCx.data_ptr(), dshared.data_ptr(), stream)
        api.unpack_shared(dshared.data_ptr(), dCx.data_ptr(), stream)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t1) / args.steps * 1e3
    t1 = time.perf_counter()
    for _ in range(args.steps):
        api.build_system_device(True, dCx.data_ptr(), drhs.data_ptr(), stream)
    torch.cuda.synchronize()
    ms_build = (time.perf_counter() - t1) / args.steps * 1e3
    api.check_status()
    free, total = torch.cuda.mem_get_info()
    print(json.dumps(dict(world=args.world, rank=args.rank, terms_total=api.n_terms, nnz=api.nnz, j_cols=api.j_cols,
                          shared_values=n_shared, synth_s=round(t_synth, 1), spec_s=round(t_spec, 1), emit_s=round(t_emit, 1),
                          setup_s=round(t_setup, 1), ms_per_step_no_collective=round(ms, 3), ms_build=round(ms_build, 3),
                          host_peak_rss_gib=round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20, 2),
                          device_used_gib=round((total - free) / 2 ** 30, 2))))


if __name__ == "__main__":
    main()
