"""Developer tool: per-phase cycle breakdown of k_brushfire (needs the -DLAMA_PROFILE_BF build of the HIP library)."""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iris_lama_amd.ffi as F
F.HIP_LIB = os.environ.get("LAMA_PROF_LIB", F.HIP_LIB)
P = int(sys.argv[1]) if len(sys.argv) > 1 else 30
pts, odom, truth = F.corridor_log(12, 1080)
ctx = F.HipContext(F.default_cfg(particles=P, profile=1, brushfire_waves=int(os.environ.get("LAMA_PROF_WAVES", "0"))))
ctx.init(pts[0], F.pose_from_xyr(*odom[0]))
L = F.hip_lib()
L.lama_hip_debug_cycles.argtypes = [C.c_void_p, C.c_void_p]
# round 4 (wave pair, straight-line lower loop): main-wave buckets of the lower wave, then the helper wave's
names = ["lookup+issue_loads", "wait_loads+decide0", "decide+commit", "general_pop", "tail(+barrier)", "H:pop", "H:wait_D", "H:pushes"]
if os.environ.get("LAMA_PROF_MAIN"):   # -DLAMA_PROFILE_BF_MAIN build: all eight buckets belong to the main wave
    names = ["lookup+issue_loads", "wait_loads+decide0", "decide+commit", "general_pop", "post_D", "pre_D", "wait_D", "before_lower(raise)"]
if os.environ.get("LAMA_PROF_FINE"):   # -DLAMA_PROFILE_BF_MAIN -DLAMA_PROFILE_BF_FINE build
    names = ["f0:lookup+issue", "f1:wait+decide0", "f2:valu_decide", "f3:rare_ballot+branch", "f4:atomic+stores", "f5:mailbox", "f6:tail_pre_D", "f7:barrier+post_D"]
if os.environ.get("LAMA_PROF_COUNT"):  # -DLAMA_PROFILE_BF_MAIN -DLAMA_PROFILE_BF_COUNT build: event counts of the lower wave
    names = ["lower_pops", "general_pops", "g:cache_miss", "g:stale", "g:alloc", "g:tie_other", "fired", "raise_pops"]
noise = float(os.environ.get("LAMA_PROF_NOISE", "0"))      # pose noise (m; rad = a third of it): particles that disagree with their maps raise as well
rng = np.random.default_rng(1)
for k in range(1, 13):
    poses = np.tile(F.pose_from_xyr(*truth[k]), (P, 1))
    if noise > 0:
        poses = np.stack([F.pose_from_xyr(*(truth[k] + rng.normal(0, [noise, noise, noise / 3]))) for _ in range(P)])
    ctx.set_poses(poses)
    ctx.reset_counters()
    ctx.update_maps(pts[k])
    c = ctx.counters()
    d = np.zeros((P, 8), dtype=np.uint64)
    L.lama_hip_debug_cycles(ctx.h, d.ctypes.data_as(C.c_void_p))
    pops = c["bf_cells"] / P
    if os.environ.get("LAMA_PROF_COUNT"):
        j = int(np.argmax(d[:, 0]))
        print(f"scan {k}: pops/particle {pops:.0f} (max lower pops {d[j][0]}) brushfire {c['ms_brushfire']:.3f} ms :: " + " ".join(f"{n}={d[j][i]}" for i, n in enumerate(names[:8])))
        continue
    if os.environ.get("LAMA_PROF_MAIN") and not os.environ.get("LAMA_PROF_FINE"):
        j = int(np.argmax(d[:, :8].sum(axis=1)))
        print(f"scan {k}: slowest particle {j}: main-wave cycles total {d[j][:8].sum()} of which before the lower wave (raise phase) {d[j][7]}  brushfire {c['ms_brushfire']:.3f} ms")
    tot = d[0][:5].sum() if not os.environ.get("LAMA_PROF_MAIN") else (d[0][:8].sum() if os.environ.get("LAMA_PROF_FINE") else d[0][:7].sum())
    print(f"scan {k}: pops/particle {pops:.0f} brushfire {c['ms_brushfire']:.3f} ms raycast {c['ms_raycast']:.3f} ms  cycles/pop {tot / max(pops,1):.0f} :: " +
          " ".join(f"{n}={d[0][i] / max(pops,1):.0f}" for i, n in enumerate(names)))
