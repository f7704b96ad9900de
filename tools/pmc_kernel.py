"""Developer tool: per-kernel means of the PMC counters in a rocprofv3 database, grouped by kernel and grid size.
python tools/pmc_kernel.py <results.db> [name-substring]"""
import sqlite3
import sys

db, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "lama_dev")
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if view is None:
    print("no counters_collection view; tables:", tabs); sys.exit(1)
cols = [r[1] for r in con.execute(f"pragma table_info({view})")]
rows = con.execute(f"select kernel_name, grid_size_x, counter_name, value from {view}").fetchall() if "grid_size_x" in cols else \
       [(a, 0, c, v) for a, c, v in con.execute(f"select kernel_name, counter_name, value from {view}")]
agg = {}
for name, gx, cn, v in rows:
    if pat not in name:
        continue
    k = (name.split("(")[0].replace("void lama_dev::", "").replace("lama_dev::", "")[:50], gx, cn)
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += v
for (name, gx, cn), (cnt, tot) in sorted(agg.items()):
    print(f"{name:50s} grid {gx:>8} {cn:24s} n {cnt:5d} mean {tot / cnt:14.1f}")
