"""Developer tool: per-phase cycle breakdown of the level-synchronous lower wave (lama_brushfire_lse.h); needs the
-DLAMA_PROFILE_LSE build of the HIP library (tools/build_prof.sh -DLAMA_PROFILE_LSE; LAMA_PROF_LIB=.../liblama_hip_prof.so)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iris_lama_amd.ffi as F
F.HIP_LIB = os.environ.get("LAMA_PROF_LIB", F.HIP_LIB)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 30
N = int(sys.argv[2]) if len(sys.argv) > 2 else 16
pts, odom, truth = F.corridor_log(N, 1080)
ctx = F.HipContext(F.default_cfg(particles=P, profile=1))
ctx.init(pts[0], F.pose_from_xyr(*odom[0]))
L = F.hip_lib()
L.lama_hip_debug_cycles.argtypes = [C.c_void_p, C.c_void_p]
L.lama_hip_debug_cycles2.argtypes = [C.c_void_p, C.c_void_p]
names = ["plan", "cells_load", "cells_resolve", "commit_cells", "commit_heap", "vevent", "serial+top"]
for k in range(1, N + 1):
    poses = np.tile(F.pose_from_xyr(*truth[k]), (P, 1))
    ctx.set_poses(poses)
    ctx.reset_counters()
    ctx.update_maps(pts[k])
    c = ctx.counters()
    d = np.zeros((P, 8), dtype=np.uint64)
    L.lama_hip_debug_cycles(ctx.h, d.ctypes.data_as(C.c_void_p))
    d2 = np.zeros((P, 8), dtype=np.uint64)
    L.lama_hip_debug_cycles2(ctx.h, d2.ctypes.data_as(C.c_void_p))
    pops = c["bf_cells"] / P
    print(f"   kernel: raise-part {d2[0][0] / 1e3:.0f}k cycles ({d2[0][2]} pops), lower wave {d2[0][1] / 1e3:.0f}k cycles (heap {d2[0][3]} at its start), spill {d2[0][4]}")
    log = np.zeros(1 << 17, dtype=np.uint64)
    L.lama_hip_debug_log.argtypes = [C.c_void_p, C.c_void_p]
    L.lama_hip_debug_log(ctx.h, log.ctypes.data_as(C.c_void_p))
    n = int(log[0])
    ev = (log[1:n] >> np.uint64(56)).astype(np.int64)
    ts = (log[1:n] & np.uint64((1 << 56) - 1)).astype(np.int64)
    dt = np.diff(ts)
    tot = float(ts[-1] - ts[0]) if n > 2 else 0.0
    sums = [int(dt[ev[1:] == i].sum()) for i in range(7)]
    cnts = [int((ev[1:] == i).sum()) for i in range(7)]
    print(f"scan {k}: pops/particle {pops:.0f} brushfire {c['ms_brushfire']:.3f} ms  lower-wave cycles {tot:.0f} events {n} :: " +
          " ".join(f"{nm}={sums[i] / 1e3:.0f}k/{cnts[i]}" for i, nm in enumerate(names)))
