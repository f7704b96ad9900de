"""Developer tool: the exact brushfire (default) against the opt-in level-synchronous variant
(cfg.brushfire_mode = 1) on the corridor log, free running, for BASELINE's three particle counts and both measurement gains:
cells whose sqdist / valid / obstacle offset differ at the end, and how far the filters' poses drift apart.
Both runs are on the device (the exact mode is what the parity tests pin to the oracle / the reference).
  python tools/canonical_report.py [steps=40]  ->  markdown table on stdout"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import iris_lama_amd.ffi as F
from _cmp import DM_FIELDS, diff_maps

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
pts, odom, truth = F.corridor_log(steps, 1080)
print("| P | gain | resamples (exact / mode 1) | max pose gap over the run [m, rad] | best-pose gap at the end [m] | particles compared | cells compared | sqdist/valid differ | obstacle offset differs | masks / patch sets / is_queued differ |")
print("|---:|---:|---|---:|---:|---:|---:|---:|---:|---:|")
for P in (30, 300, 3000):
    for gain in (3.0, 0.01):
        a = F.PFSlam2D(F.pf_options(particles=P, seed=42, meas_sigma_gain=gain, brushfire_mode=0))
        b = F.PFSlam2D(F.pf_options(particles=P, seed=42, meas_sigma_gain=gain, brushfire_mode=1))
        a.set_prior(*odom[0]); b.set_prior(*odom[0])
        gap = 0.0
        for k in range(steps + 1):
            a.update(pts[k], odom[k], float(k)); b.update(pts[k], odom[k], float(k))
            gap = max(gap, float(np.abs(a.poses() - b.poses()).max()))
        best = float(np.abs(np.array(a.best_pose_xyr()) - np.array(b.best_pose_xyr()))[:2].max())
        ca, cb = a.hip_context(), b.hip_context()
        sample = range(P) if P <= 30 else np.linspace(0, P - 1, 30).astype(int)
        tot = dv = ob = other = 0
        for i in sample:
            da, db = ca.download_map(int(i), F.MAP_DISTANCE), cb.download_map(int(i), F.MAP_DISTANCE)
            d = diff_maps(da, db, DM_FIELDS)
            tot += 1024 * len(da)
            dv += max(d["sqdist"], d["valid"]); ob += d["obstacle"]
            other += d["mask_words"] + d["patches_only_dev"] + d["patches_only_orc"] + d["queued"]
        print(f"| {P} | {gain} | {a.num_resamples()} / {b.num_resamples()} | {gap:.3g} | {best:.3g} | {len(list(sample))} | {tot} | {dv} | {ob} | {other} |", flush=True)
        a.close(); b.close()
