"""Developer tool: cycle breakdown of the brushfire HELPER wave (needs a -DLAMA_PROFILE_BF -DLAMA_PROFILE_BFH build)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iris_lama_amd.ffi as F
F.HIP_LIB = os.environ.get("LAMA_PROF_LIB", F.HIP_LIB)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 30
pts, odom, truth = F.corridor_log(12, 1080)
ctx = F.HipContext(F.default_cfg(particles=P, profile=1))
ctx.init(pts[0], F.pose_from_xyr(*odom[0]))
L = F.hip_lib()
L.lama_hip_debug_cycles.argtypes = [C.c_void_p, C.c_void_p]
names = ["-", "wait_D", "pushes", "begin", "chunks", "finish", "topq", "n_chunks"]
for k in range(1, 13):
    poses = np.tile(F.pose_from_xyr(*truth[k]), (P, 1))
    ctx.set_poses(poses)
    ctx.reset_counters()
    ctx.update_maps(pts[k])
    c = ctx.counters()
    d = np.zeros((P, 8), dtype=np.uint64)
    L.lama_hip_debug_cycles(ctx.h, d.ctypes.data_as(C.c_void_p))
    pops = c["bf_cells"] / P
    print(f"scan {k}: pops/particle {pops:.0f} brushfire {c['ms_brushfire']:.3f} ms :: " +
          " ".join(f"{n}={d[0][i] / max(pops,1):.2f}" for i, n in enumerate(names)))
