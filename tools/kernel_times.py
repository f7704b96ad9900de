"""Developer tool: per-kernel mean durations out of a rocprofv3 --kernel-trace database, grouped by kernel and grid size (so that
the particle counts of a bench sweep stay apart).  python tools/kernel_times.py <results.db> [name-substring]"""
import sqlite3
import sys

db, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "lama_dev")
rows = sqlite3.connect(db).cursor().execute("select name, grid_x, grid_y, duration from kernels").fetchall()
agg = {}
for name, gx, gy, dur in rows:
    if pat not in name:
        continue
    k = (name.split("(")[0].replace("void lama_dev::", "").replace("lama_dev::", ""), gx, gy)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += dur
for (name, gx, gy), (cnt, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{name[:60]:60s} grid {gx:>8}x{gy:<5} calls {cnt:5d}  mean {tot / cnt / 1e3:9.2f} us  total {tot / 1e6:8.2f} ms")
