"""Developer tool (needs the -DLAMA_PROFILE_BF -DLAMA_PROFILE_BF_MAIN build): when and where each particle's first-stage brushfire
workgroup ran -- start / end on the constant 100 MHz counter, XCD, CU -- and the spread of the per-particle cycle totals.
usage: LAMA_PROF_LIB=tools/_prof/liblama_hip_prof_main.so python tools/prof_bf_spread.py [P]"""
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
for f in (L.lama_hip_debug_cycles, L.lama_hip_debug_cycles2):
    f.argtypes = [C.c_void_p, C.c_void_p]
for k in range(1, 13):
    ctx.set_poses(np.tile(F.pose_from_xyr(*truth[k]), (P, 1)))
    ctx.reset_counters()
    ctx.update_maps(pts[k])
    c = ctx.counters()
    if k < 10:
        continue
    d = np.zeros((P, 8), dtype=np.uint64); e = np.zeros((P, 8), dtype=np.uint64)
    L.lama_hip_debug_cycles(ctx.h, d.ctypes.data_as(C.c_void_p))
    L.lama_hip_debug_cycles2(ctx.h, e.ctypes.data_as(C.c_void_p))
    cyc = d.sum(axis=1).astype(np.float64)
    t0 = e[:, 0].astype(np.float64); t1 = e[:, 1].astype(np.float64)
    base = t0.min()
    start_us = (t0 - base) / 100.0; end_us = (t1 - base) / 100.0
    dur_us = end_us - start_us
    ghz = cyc / (dur_us * 1e3)
    xcc = (e[:, 3] & 0xF).astype(int)
    cu = ((e[:, 2] >> 8) & 0xF).astype(int); se = ((e[:, 2] >> 13) & 0x7).astype(int); simd = ((e[:, 2] >> 4) & 0x3).astype(int)
    print(f"scan {k}: brushfire {c['ms_brushfire']:.3f} ms (all stages); first-stage span {end_us.max() / 1e3:.3f} ms")
    print(f"   start  us: min {start_us.min():.0f} median {np.median(start_us):.0f} p90 {np.percentile(start_us, 90):.0f} max {start_us.max():.0f}")
    print(f"   dur    us: min {dur_us.min():.0f} median {np.median(dur_us):.0f} p90 {np.percentile(dur_us, 90):.0f} max {dur_us.max():.0f}")
    print(f"   cycles   : min {cyc.min():.0f} median {np.median(cyc):.0f} max {cyc.max():.0f};  cycles/us: median {np.median(ghz):.2f} min {ghz.min():.2f} max {ghz.max():.2f} (the kernel's own timers cover the loops only)")
    per = {}
    for x, s_, c_ in zip(xcc, se, cu):
        per[(x, s_, c_)] = per.get((x, s_, c_), 0) + 1
    occ = np.array(list(per.values()))
    print(f"   workgroups per CU: {len(per)} CUs used, min {occ.min()} median {np.median(occ):.0f} max {occ.max()};  per XCD: {np.bincount(xcc, minlength=8).tolist()}")
    late = start_us > 50
    print(f"   late starters (> 50 us): {late.sum()};  dur of the slowest 1%: {np.sort(dur_us)[-max(P // 100, 1):].mean():.0f} us")
    # main waves per SIMD against the duration: is it the placement?
    sh_ = ((e[:, 2] >> 12) & 0x1).astype(int)
    key = xcc * 100000 + se * 10000 + sh_ * 1000 + cu * 10 + simd
    uniq, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    n_on_simd = cnt[inv]
    for n in sorted(set(n_on_simd.tolist())):
        m = n_on_simd == n
        print(f"   main waves sharing the SIMD = {n}: {m.sum()} particles, duration median {np.median(dur_us[m]):.0f} us (min {dur_us[m].min():.0f}, max {dur_us[m].max():.0f})")
    print(f"   SIMD ids of the main waves: {np.bincount(simd, minlength=4).tolist()}")
