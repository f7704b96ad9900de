#!/usr/bin/env python
"""Turn the rocprofv3 (ROCm 7.2, rocpd SQLite) outputs of tools/profile_round.sh into the committed summaries:
  profiles/<tag>_rocprof_stats.md      per-kernel time (all dispatches, and the last K = timed-region dispatches)
  profiles/pmc_brushfire.json          HBM traffic per launch of the dominant kernel (read by bench.py)
FETCH_SIZE / WRITE_SIZE are in KiB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts 128-B requests
as 64 B for wide streaming reads, so the read side is doubled ("corrected"); for the narrow scattered accesses of these
kernels the factor is uncalibrated, hence both raw and corrected values are recorded."""
import json
import os
import sqlite3
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r01"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
K = int(sys.argv[3]) if len(sys.argv) > 3 else 30          # dispatches of the timed region (bench --steps)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def q(db, sql):
    return sqlite3.connect(db).cursor().execute(sql).fetchall()


out = [f"# rocprofv3 summary ({tag})", "",
       f"Command: `rocprofv3 --kernel-trace --stats -- python bench.py --steps {K} --warmup 5 --no-cpu --sweep ''` "
       "(MI355X, gfx950, ROCm 7.2).", "",
       f"## All dispatches (rocprofv3 `top_kernels` view: first scan + 5 warm-up + {K} timed updates; bench.py runs the pass four times: value, kernel brackets, forced resampling, Summary)", "",
       "| kernel | calls | total ms | avg ms | % |", "|---|---:|---:|---:|---:|"]
tdb = os.path.join(src, "trace", f"{tag}_results.db")
for name, calls, tot, avg, pct in q(tdb, "select name,total_calls,total_duration,average,percentage from top_kernels"):
    out.append(f"| `{name.split('(')[0]}` | {calls} | {tot / 1e3:.1f} | {avg / 1e3:.2f} | {pct:.2f} |")
out += ["", f"## Timed region only (last {K} dispatches of each kernel = what bench.py brackets with hipEvents)", "",
        "| kernel | dispatches | avg us | min us | max us | grid | wg | LDS B | VGPR | SGPR |", "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
rows = q(tdb, "select name,duration,grid_x,workgroup_x,lds_size,vgpr_count,sgpr_count,start from kernels order by start")
by = {}
for r in rows:
    by.setdefault(r[0], []).append(r)
timed = {}
for name, lst in by.items():
    if "lama_dev" not in name:
        continue
    last = lst[-K:] if len(lst) >= K else lst
    d = [x[1] / 1e3 for x in last]
    timed[name] = sum(d) / len(d)
    out.append(f"| `{name.split('(')[0]}` | {len(last)} | {sum(d) / len(d):.2f} | {min(d):.2f} | {max(d):.2f} | {last[-1][2]} | {last[-1][3]} | {last[-1][4]} | {last[-1][5]} | {last[-1][6]} |")


def pmc(dbdir, counter):
    db = os.path.join(src, dbdir, f"{tag}_results.db")
    if not os.path.exists(db):
        return {}
    rows = q(db, f"select kernel_name,value,start from counters_collection where counter_name='{counter}' order by start")
    by = {}
    for n, v, s in rows:
        by.setdefault(n, []).append(v)
    return {n: (sum(v[-K:]) / len(v[-K:])) for n, v in by.items() if "lama_dev" in n}


fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
out += ["", f"## HBM traffic per launch (PMC, separate passes; mean over the last {K} dispatches)", "",
        "| kernel | FETCH_SIZE KiB | WRITE_SIZE KiB | raw bytes | corrected bytes (2x read) |", "|---|---:|---:|---:|---:|"]
pj = {}
for name in sorted(set(fetch) | set(write)):
    f, w = fetch.get(name, 0.0), write.get(name, 0.0)
    raw, cor = (f + w) * 1024, (2 * f + w) * 1024
    out.append(f"| `{name.split('(')[0]}` | {f:.1f} | {w:.1f} | {raw:.0f} | {cor:.0f} |")
    if "k_brushfire<1024" in name:
        pj = {"kernel": name.split("(")[0], "fetch_kib": f, "write_kib": w, "hbm_bytes_per_launch_raw": raw,
              "hbm_bytes_per_launch": cor, "mean_launch_us_timed_region": timed.get(name),
              "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950); uncalibrated for narrow scattered accesses"}
sq = {}
sdb = os.path.join(src, "pmc_sq", f"{tag}_results.db")
if os.path.exists(sdb):
    rows = q(sdb, "select kernel_name,counter_name,value,start from counters_collection order by start")
    acc = {}
    for n, cn, v, s in rows:
        if "lama_dev" in n:
            acc.setdefault((n, cn), []).append(v)
    names = sorted({k[0] for k in acc})
    ctrs = sorted({k[1] for k in acc})
    out += ["", f"## SQ counters per launch (mean over the last {K} dispatches)", "", "| kernel | " + " | ".join(ctrs) + " |",
            "|---|" + "---:|" * len(ctrs)]
    for n in names:
        vals = []
        for cn in ctrs:
            v = acc.get((n, cn), [0])[-K:]
            vals.append(sum(v) / len(v))
        out.append(f"| `{n.split('(')[0]}` | " + " | ".join(f"{v:.3g}" for v in vals) + " |")
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
open(os.path.join(ROOT, "profiles", f"{tag}_rocprof_stats.md"), "w").write("\n".join(out) + "\n")
if pj:
    import subprocess, datetime
    try:
        pj["git_head"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except Exception:
        pj["git_head"] = None
    pj["collected"] = datetime.date.today().isoformat() + " (" + tag + ")"
    json.dump(pj, open(os.path.join(ROOT, "profiles", "pmc_brushfire.json"), "w"), indent=1)
print("\n".join(out))
