import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import iris_lama_amd.ffi as F
gpus, P, gain = 8, 3000, 1e-4
steps = 6
pts, odom, _ = F.corridor_log(steps, 1080)
for trial in range(int(os.environ.get('TRIALS', '30'))):
    a = F.PFSlam2D(F.pf_options(particles=P, seed=42, meas_sigma_gain=gain))
    b = F.PFSlam2D(F.pf_options(particles=P, seed=42, meas_sigma_gain=gain, gpus=gpus))
    a.set_prior(*odom[0]); b.set_prior(*odom[0])
    ok = True
    for k in range(steps + 1):
        a.update(pts[k], odom[k], float(k)); b.update(pts[k], odom[k], float(k))
        pe = np.array_equal(a.poses(), b.poses())
        wa, wb = a.weights(), b.weights()
        we = [np.array_equal(u, v) for u, v in zip(wa, wb)]
        ca = a.hip_context()
        cks = {}
        for kind in (F.MAP_DISTANCE, F.MAP_OCCUPANCY):
            cb = np.concatenate([b.shard_context(r).map_checksums(kind) for r in range(gpus)])
            cks[kind] = np.nonzero(cb != ca.map_checksums(kind))[0]
        if pe and all(we) and not len(cks[F.MAP_DISTANCE]) and not len(cks[F.MAP_OCCUPANCY]): continue
        print("trial", trial, "step", k, "poses", pe, "weights", we, "resamples", a.num_resamples(), b.num_resamples(),
              "dm diff", cks[F.MAP_DISTANCE][:8], len(cks[F.MAP_DISTANCE]), "occ diff", cks[F.MAP_OCCUPANCY][:8], len(cks[F.MAP_OCCUPANCY]), flush=True)
        if not all(we):
            for u, v in zip(wa, wb):
                d = np.nonzero(u != v)[0]
                print("   differing entries", len(d), d[:10], (u[d[:5]], v[d[:5]]))
                if len(d): print("   max rel diff", np.max(np.abs(u[d] - v[d]) / np.maximum(np.abs(u[d]), 1e-300)), "argmax", d[np.argmax(np.abs(u[d] - v[d]) / np.maximum(np.abs(u[d]), 1e-300))])
            ok = False
            break
    a.close(); b.close()
    print("trial", trial, "ok" if ok else "MISMATCH", flush=True)
