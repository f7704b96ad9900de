"""Developer tool: brushfire time per scan for the wave layouts of the exact brushfire (cfg.brushfire_waves = 1 / 2 / 3) at several
particle counts, teacher-forced corridor scans (same work for every layout).  python tools/bf_waves_sweep.py [P ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iris_lama_amd.ffi as F
Ps = [int(a) for a in sys.argv[1:]] or [30, 300, 1000, 3000]
pts, odom, truth = F.corridor_log(14, 1080)
for P in Ps:
    row = []
    for waves in (1, 2):
        ctx = F.HipContext(F.default_cfg(particles=P, profile=1, brushfire_waves=waves))
        ctx.init(pts[0], F.pose_from_xyr(*odom[0]))
        rng = np.random.default_rng(1)
        for k in range(1, 15):
            base = F.pose_from_xyr(*truth[k])
            poses = np.tile(base, (P, 1))
            poses[:, 2:] += rng.normal(0, 0.01, (P, 2))
            ctx.set_poses(poses)
            if k == 5:
                ctx.reset_counters()
            ctx.update_maps(pts[k])
        c = ctx.counters()
        row.append((waves, c["ms_brushfire"] / max(c["launches_brushfire"], 1), c["ms_raycast"] / max(c["launches_raycast"], 1), c["bf_cells"] / P / max(c["launches_brushfire"], 1)))
        ctx.close()
    print(f"P={P}: " + "  ".join(f"waves={w}: brushfire {b:.3f} ms (raycast {r:.3f}, pops/particle-scan {n:.0f})" for w, b, r, n in row), flush=True)
