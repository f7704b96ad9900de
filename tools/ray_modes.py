"""Developer tool: ray-cast kernel time per scan for the beam-sequential and the parallel form at a given particle count."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iris_lama_amd.ffi as F
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
pts, odom, truth = F.corridor_log(8, 1080)
for mode in (1, 2):
    ctx = F.HipContext(F.default_cfg(particles=P, profile=1, sequential_raycast=mode))
    ctx.init(pts[0], F.pose_from_xyr(*odom[0]))
    rng = np.random.default_rng(0)
    tot = 0.0
    for k in range(1, 9):
        poses = np.stack([F.pose_from_xyr(*(truth[k] + rng.normal(0, [0.03, 0.03, 0.01]))) for _ in range(P)])
        ctx.set_poses(poses)
        ctx.reset_counters()
        ctx.update_maps(pts[k])
        c = ctx.counters()
        if k > 2:
            tot += c["ms_raycast"]
    print(f"P={P} sequential_raycast={mode}: raycast {tot / 6:.3f} ms/scan, brushfire {c['ms_brushfire']:.3f} ms (last)")
    ctx.close()
