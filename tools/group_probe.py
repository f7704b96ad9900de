"""One lama::PFSlam2D object with Options::gpus = G (all shards on the devices that exist): ms per update and where it goes.
usage: python tools/group_probe.py [P] [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iris_lama_amd.ffi as F

P = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warm = 5
pts, odom, _ = F.corridor_log(steps + warm + 1, 1080)
for G in (1, 2, 4, 8):
    pf = F.PFSlam2D(F.pf_options(particles=P, seed=42, gpu_device=0, gpus=G, create_summary=0))
    pf.set_prior(*odom[0])
    for k in range(warm + 1):
        pf.update(pts[k], odom[k], float(k))
    for r in range(G):
        (pf.shard_context(r) if G > 1 else pf.hip_context()).sync()
    t0 = time.perf_counter(); xs = []
    for k in range(warm + 1, warm + steps + 1):
        pf.update(pts[k], odom[k], float(k)); xs.append(pf.exchange_times())
    for r in range(G):
        (pf.shard_context(r) if G > 1 else pf.hip_context()).sync()
    dt = (time.perf_counter() - t0) / steps
    print(f"G={G} P={P}: {1e3*dt:.2f} ms/update  begin {1e3*np.mean([x['phase_begin_s'] for x in xs]):.2f}  maps(launch+mirror) {1e3*np.mean([x['phase_maps_s'] for x in xs]):.2f}", flush=True)
    pf.close()
