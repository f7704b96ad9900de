"""Developer tool (needs the -DLAMA_PROFILE_RAY build): what k_ray_patches does per particle-scan -- patches, chunks kept, beams
tested at each stage, crossings, cells -- and where thread 0 of its workgroups spends its cycles.
usage: LAMA_PROF_LIB=tools/_prof/liblama_hip_prof_ray.so python tools/prof_ray.py [P]"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iris_lama_amd.ffi as F
F.HIP_LIB = os.environ.get("LAMA_PROF_LIB", F.HIP_LIB)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
pts, odom, truth = F.corridor_log(12, 1080)
ctx = F.HipContext(F.default_cfg(particles=P, profile=1))
ctx.init(pts[0], F.pose_from_xyr(*odom[0]))
L = F.hip_lib()
L.lama_hip_debug_log.argtypes = [C.c_void_p, C.c_void_p]
names = ["patches", "patches_with_active", "chunks_kept", "beams_tested", "box_pass", "line_pass", "crossings", "cells",
         "cyc_classify", "cyc_chunks", "cyc_beams", "cyc_walk", "cyc_writeback"]
def read():
    d = np.zeros(1 << 17, dtype=np.uint64)
    L.lama_hip_debug_log(ctx.h, d.ctypes.data_as(C.c_void_p))
    return d
GY = 128 if P <= 64 else 32
NP = min(P, 64)
rng = np.random.default_rng(0)
for k in range(1, 13):
    poses = np.stack([F.pose_from_xyr(*(np.asarray(truth[k]) + rng.normal(0, [0.03, 0.03, 0.01]))) for _ in range(P)])
    ctx.set_poses(poses)
    ctx.reset_counters()
    a = read()[:16].astype(np.float64)
    ctx.update_maps(pts[k])
    c = ctx.counters()
    raw = read()
    d = (raw[:16].astype(np.float64) - a) / P
    if k >= 10:
        print(f"scan {k}: raycast {c['ms_raycast']:.3f} ms; per particle-scan: " + ", ".join(f"{n} {d[i]:.0f}" for i, n in enumerate(names[:8])))
        t = raw[16:16 + 8 * NP * GY].reshape(NP * GY, 8).astype(np.float64)
        patches = t[:, 5].sum()
        tot = t[:, :5].sum()
        print(f"          thread 0, first {NP} particles: {patches / NP:.0f} patches per particle, {tot / max(patches, 1):.0f} cycles per patch: " +
              ", ".join(f"{n} {t[:, i].sum() / max(patches, 1):.0f}" for i, n in enumerate(["loads+classify", "-", "tests", "walk", "writeback"])))
        busy = t[:, :5].sum(axis=1)
        print(f"          per workgroup: cycles median {np.median(busy[busy > 0]):.0f} max {busy.max():.0f}; patches per workgroup max {t[:, 5].max():.0f}")
