"""Developer tool: scan-match launch time per Gauss-Newton evaluation on the bench log."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import iris_lama_amd.ffi as F
F.HIP_LIB = os.environ.get("LAMA_PROF_LIB", F.HIP_LIB)      # a -DLAMA_PROFILE_SM build adds the cycle breakdown
P = int(sys.argv[1]) if len(sys.argv) > 1 else 30
pts, odom, truth = F.corridor_log(12, 1080)
ctx = F.HipContext(F.default_cfg(particles=P, profile=1))
ctx.init(pts[0], F.pose_from_xyr(*odom[0]))
rng = np.random.default_rng(0)
for k in range(1, 13):
    noise = rng.normal(0, [0.03, 0.03, 0.01], size=(P, 3))
    poses = np.stack([F.pose_from_xyr(*(np.array(truth[k]) + noise[i])) for i in range(P)])
    ctx.set_poses(poses)
    ctx.reset_counters()
    g_poses, ll, it = ctx.scan_match(pts[k])
    c = ctx.counters()
    ev = c["gn_evals"] / P
    print(f"scan {k}: scan_match {c['ms_scan_match']*1e3:.1f} us, iterations mean {it.mean():.1f} max {it.max()}, evals/particle {ev:.1f}, "
          f"us per max-iteration {c['ms_scan_match']*1e3/max(it.max(),1):.2f}")
    if "LAMA_PROF_LIB" in os.environ:
        L = F.hip_lib()
        L.lama_hip_debug_cycles.argtypes = [C.c_void_p, C.c_void_p]
        d = np.zeros((P, 8), dtype=np.uint64)
        L.lama_hip_debug_cycles(ctx.h, d.ctypes.data_as(C.c_void_p))
        j = int(np.argmax(it))
        names = ["eval", "block_sum", "step(thread 0)+sync", "validate(thread 0)+sync"]
        print("   slowest particle, cycles per iteration: " + " ".join(f"{n}={d[j][i] / max(it[j], 1):.0f}" for i, n in enumerate(names)))
    ctx.set_poses(g_poses)
    ctx.update_maps(pts[k])
