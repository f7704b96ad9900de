"""Worker for the world_size>1 tests: runs the sharded PFSlam2D driver on one rank and dumps what it owns."""
import os
import pickle
import sys

import numpy as np


def run(rank, world, port, backend, engine_path, P, steps, beams, gain, out_dir, gpu, checksums_only=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    sys.path.insert(0, here)
    import torch
    import torch.distributed as dist
    import iris_lama_amd.ffi as F
    from iris_lama_amd.distributed import ShardedPF

    if backend == "nccl":                # one process per GPU: RCCL binds the process' current device
        torch.cuda.set_device(gpu)
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    # the process group works (one collective through it): from here on a failure is the code's, not the box's
    probe = torch.zeros(1, device=torch.device("cuda", gpu) if backend == "nccl" else torch.device("cpu"))
    dist.all_reduce(probe)
    open(os.path.join(out_dir, f"rank{rank}.up"), "w").close()
    if engine_path:
        import _testhost                 # test double of the device C-ABI: needs the test-suite's own -DLAMA_TESTING host build
        _testhost.set_engine_library(engine_path)
    pts, odom, _ = F.corridor_log(steps, beams)
    more = {"l2_max": float(os.environ["LAMA_TEST_L2_MAX"])} if os.environ.get("LAMA_TEST_L2_MAX") else {}     # (the wide device library)
    opts = F.pf_options(particles=P, seed=42, meas_sigma_gain=gain, shard_rank=rank, shard_world=world, gpu_device=gpu, **more)
    pf = ShardedPF(opts, device=torch.device("cpu") if backend == "gloo" else None)
    pf.set_prior(*odom[0])
    hist = []
    for k in range(steps + 1):
        ok = pf.update(pts[k], odom[k], float(k))
        w, nw, ws = pf.pf.weights()
        hist.append(dict(ok=ok, poses=pf.pf.poses()[pf.pf.lo:pf.pf.hi].copy(), w=w.copy(), ws=ws.copy(), neff=pf.pf.neff(),
                         best=pf.pf.best()))
    ctx = pf.pf.hip_context()
    maps, sums = {}, None
    if checksums_only:        # large pools: one device-side checksum per particle and map instead of the maps themselves
        sums = (ctx.map_checksums(F.MAP_DISTANCE), ctx.map_checksums(F.MAP_OCCUPANCY))
    else:
        for i in range(pf.pf.lo, pf.pf.hi):
            maps[i] = (ctx.download_map(i - pf.pf.lo, F.MAP_DISTANCE), ctx.download_map(i - pf.pf.lo, F.MAP_OCCUPANCY))
    res = dict(lo=pf.pf.lo, hi=pf.pf.hi, hist=hist, maps=maps, sums=sums, resamples=pf.pf.num_resamples(), shipped=pf.shipped_particles,
               origin=pf.pf.engine_origin(), device=ctx.device(), backend=pf.backend)
    with open(os.path.join(out_dir, f"rank{rank}.pkl"), "wb") as f:
        pickle.dump(res, f)
    dist.barrier()
    pf.close()
    dist.destroy_process_group()
