"""Developer tool: per-update spread of the brushfire chains over the pool (LAMA_HIP_DEBUG_TAIL=1 prints on stderr)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["LAMA_HIP_DEBUG_TAIL"] = "1"
import numpy as np
import iris_lama_amd.ffi as F
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
pts, odom, _ = F.corridor_log(steps, 1080)
pf = F.PFSlam2D(F.pf_options(particles=P, seed=42, profile=1))
pf.set_prior(*odom[0])
for k in range(steps + 1):
    pf.update(pts[k], odom[k], float(k))
    c = pf.hip_context().counters()
    print(k, "ms_brushfire(cum)", round(c["ms_brushfire"], 3), "handovers", c["brushfire_handovers"], flush=True)
