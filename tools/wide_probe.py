#!/usr/bin/env python
"""What a large l2_max costs on the device (run on the MI355X box): lama::PFSlam2D on the corridor log with the reach of the distance
map at 0.5 m (the default, liblama_hip.so), 3 m, 7 m and 12.75 m (the last two: liblama_hip_wide.so), 30 particles, 12 scans after 3
warm-up; the oracle (one thread) beside it for the pops and the CPU time.  Prints ms per update and the brushfire counters."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import iris_lama_amd.ffi as F      # noqa: E402

P, WARM, STEPS = 30, 3, 12
pts, odom, truth = F.corridor_log(WARM + STEPS, 1080)
for l2 in (0.5, 3.0, 7.0, 12.75):
    pf = F.PFSlam2D(F.pf_options(particles=P, seed=42, l2_max=l2, create_summary=0, **({} if "--default-queue" in sys.argv else {"queue_capacity": 1 << 20})))
    pf.set_prior(*odom[0])
    t0 = None
    for k in range(WARM + STEPS + 1):
        if k == WARM + 1:
            pf.hip_context().reset_counters()
            t0 = time.perf_counter()
        pf.update(pts[k], odom[k], float(k))
    dt = (time.perf_counter() - t0) / STEPS
    c = pf.hip_context().counters()
    keep = {k: c[k] for k in ("ms_brushfire", "launches_brushfire", "brushfire_handovers", "brushfire_routed", "bf_cells", "bf_longest_chain_sum", "hbm_bytes_used") if k in c}
    print(f"l2_max {l2:5.2f} m ({os.path.basename(pf.engine_origin())}): {dt * 1e3:8.2f} ms per update, {P / dt:9.0f} particle-scans/s  {keep}", flush=True)
    pf.close()
    if "--cpu" in sys.argv:
        import _oracle as O
        o = O.PF(O.default_options(particles=P, seed=42, l2_max=l2))
        o.set_prior(O.se2(*odom[0]))
        for k in range(WARM + STEPS + 1):
            if k == WARM + 1:
                t0 = time.perf_counter()
            o.update(pts[k], O.se2(*odom[k]), float(k))
        dt = (time.perf_counter() - t0) / STEPS
        print(f"             CPU oracle, one thread: {dt * 1e3:8.2f} ms per update, {P / dt:9.0f} particle-scans/s", flush=True)
