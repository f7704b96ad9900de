"""Developer tool: where a PFSlam2D::update step spends its wall time (host buckets vs device kernel time)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iris_lama_amd.ffi as F
P = int(sys.argv[1]) if len(sys.argv) > 1 else 30
prof = int(sys.argv[2]) if len(sys.argv) > 2 else 1
pts, odom, truth = F.corridor_log(35, 1080)
pf = F.PFSlam2D(F.pf_options(particles=P, seed=42, create_summary=1, profile=prof))
pf.set_prior(*odom[0])
pf.update(pts[0], odom[0], 0.0)
for k in range(1, 6):
    pf.update(pts[k], odom[k], float(k))
ctx = pf.hip_context()
ctx.reset_counters()
acc = dict(total=0.0, solving=0.0, normalizing=0.0, resampling=0.0, mapping=0.0)
t0 = time.perf_counter()
for k in range(6, 36):
    pf.update(pts[k], odom[k], float(k))
    t = pf.last_times()
    for n in acc:
        acc[n] += t[n]
wall = (time.perf_counter() - t0) / 30 * 1e3
c = ctx.counters()
print(f"P={P} profile={prof}: wall/step {wall:.3f} ms (python loop); host buckets per step [ms]: " + ", ".join(f"{n}={1e3 * v / 30:.3f}" for n, v in acc.items()))
if prof:
    print(f"  device: scan_match {c['ms_scan_match'] / 30:.3f}, raycast {c['ms_raycast'] / 30:.3f}, brushfire {c['ms_brushfire'] / 30:.3f} ms")
