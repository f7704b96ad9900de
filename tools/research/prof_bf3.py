"""Developer tool (needs the -DLAMA_PROFILE_BF3 build): where main wave A and the helper wave of the three-wave brushfire spend their
cycles in the lower phase.  (for the kernel of tools/research/brushfire_three_waves.patch) usage: LAMA_PROF_LIB=... python tools/research/prof_bf3.py [P]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import iris_lama_amd.ffi as F
F.HIP_LIB = os.environ.get("LAMA_PROF_LIB", F.HIP_LIB)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 30
pts, odom, truth = F.corridor_log(12, 1080)
ctx = F.HipContext(F.default_cfg(particles=P, profile=1, brushfire_waves=3))
ctx.init(pts[0], F.pose_from_xyr(*odom[0]))
L = F.hip_lib()
for f in (L.lama_hip_debug_cycles, L.lama_hip_debug_cycles2):
    f.argtypes = [C.c_void_p, C.c_void_p]
for k in range(1, 13):
    ctx.set_poses(np.tile(F.pose_from_xyr(*truth[k]), (P, 1)))
    ctx.reset_counters()
    ctx.update_maps(pts[k])
    c = ctx.counters()
    if k < 10:
        continue
    a = np.zeros((P, 8), dtype=np.uint64); h = np.zeros((P, 8), dtype=np.uint64)
    L.lama_hip_debug_cycles(ctx.h, a.ctypes.data_as(C.c_void_p))
    L.lama_hip_debug_cycles2(ctx.h, h.ctypes.data_as(C.c_void_p))
    a = a[0].astype(float); h = h[0].astype(float)
    na, nh = max(a[7], 1), max(h[3], 1)
    print(f"scan {k}: brushfire {c['ms_brushfire']:.3f} ms, pops/particle {c['bf_cells'] / P:.0f}; wave A: {na:.0f} pops, per pop: wait_pop {a[0]/na:.0f} wait_drain {a[1]/na:.0f} "
          f"hint_loads+D1 {a[6]/na:.0f} loads+D1 {a[2]/na:.0f} wait_dec {a[3]/na:.0f} validate {a[4]/na:.0f} D2+post {a[5]/na:.0f} = {a[:7].sum()/na:.0f} per own pop; "
          f"[entry displaced {h[4]:.0f}, forwarded {h[5]:.0f}, hint wrong {h[6]:.0f}] helper: {nh:.0f} pops, per pop: pop {h[0]/nh:.0f} wait_dec {h[1]/nh:.0f} push {h[2]/nh:.0f} = {h[:3].sum()/nh:.0f}")
