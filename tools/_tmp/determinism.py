import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import iris_lama_amd.ffi as F
P, gain, steps = 3000, 1e-4, 6
gpus = int(os.environ.get("GPUS", "1"))
pts, odom, _ = F.corridor_log(steps, 1080)
base = None
N = int(os.environ.get("TRIALS", "100"))
bad = 0
for trial in range(N):
    kw = dict(gpus=gpus) if gpus > 1 else {}
    a = F.PFSlam2D(F.pf_options(particles=P, seed=42, meas_sigma_gain=gain, **kw))
    a.set_prior(*odom[0])
    rec = []
    for k in range(steps + 1):
        a.update(pts[k], odom[k], float(k))
        if gpus > 1:
            dm = np.concatenate([a.shard_context(r).map_checksums(F.MAP_DISTANCE) for r in range(gpus)])
            oc = np.concatenate([a.shard_context(r).map_checksums(F.MAP_OCCUPANCY) for r in range(gpus)])
            c = a.shard_context(0).counters()
        else:
            ctx = a.hip_context()
            dm, oc = ctx.map_checksums(F.MAP_DISTANCE), ctx.map_checksums(F.MAP_OCCUPANCY)
            c = ctx.counters()
        rec.append((a.poses().copy(), a.weights()[1].copy(), dm, oc, (c["brushfire_routed"], c["brushfire_early"], c["brushfire_handovers"], c["replay_handovers"])))
    a.close()
    if base is None:
        base = rec
        print("baseline counters per step", [r[4] for r in rec], flush=True)
        continue
    for k in range(steps + 1):
        names = ("poses", "nweights", "dm", "occ")
        diff = [n for n, x, y in zip(names, rec[k][:4], base[k][:4]) if not np.array_equal(x, y)]
        if diff:
            bad += 1
            d_dm = np.nonzero(rec[k][2] != base[k][2])[0]; d_oc = np.nonzero(rec[k][3] != base[k][3])[0]
            d_p = np.nonzero((rec[k][0] != base[k][0]).any(axis=1))[0]
            print("trial", trial, "FIRST DIVERGENCE at step", k, diff, "dm particles", d_dm[:10], "occ particles", d_oc[:10], "pose particles", d_p[:10],
                  "counters", rec[k][4], "baseline", base[k][4], flush=True)
            break
print("trials", N, "divergent", bad)
