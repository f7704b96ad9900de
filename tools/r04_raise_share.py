"""Developer tool (COUNT profiling build): raise / lower pops per particle over the bench's free-running filter."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import iris_lama_amd.ffi as F
F.HIP_LIB = os.environ.get("LAMA_PROF_LIB", "tools/_prof/liblama_hip_prof_count.so")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 30
steps = 25
pts, odom, _ = F.corridor_log(steps, 1080)
pf = F.PFSlam2D(F.pf_options(particles=P, seed=42, profile=1))
pf.set_prior(*odom[0])
L = F.hip_lib()
L.lama_hip_debug_cycles.argtypes = [C.c_void_p, C.c_void_p]
prev = 0.0
for k in range(steps + 1):
    pf.update(pts[k], odom[k], float(k))
    ctx = pf.hip_context()
    c = ctx.counters()
    d = np.zeros((P, 8), dtype=np.uint64)
    L.lama_hip_debug_cycles(ctx.h, d.ctypes.data_as(C.c_void_p))
    tot = d[:, 0] + d[:, 7]
    j = int(np.argmax(tot))
    print(f"step {k}: brushfire {c['ms_brushfire'] - prev:.3f} ms  longest: lower {d[j][0]} raise {d[j][7]}  pool mean: lower {d[:,0].mean():.0f} raise {d[:,7].mean():.0f}  max raise {d[:,7].max()}", flush=True)
    prev = c["ms_brushfire"]
