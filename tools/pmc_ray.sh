#!/bin/bash
# Developer tool (every profiler pass under its own `timeout`: an unknown counter name makes rocprofv3 abort and hang): SQ / memory counters of the ray-cast kernels at a given particle count (two PMC passes, kernel trace only).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
P=${1:-3000}
OUT=gpurun_out/pmc_ray_$P; mkdir -p $OUT
CMD="python bench.py --no-cpu --particles $P --steps 6 --warmup 3 --sweep="
timeout 150 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES -d $OUT/sq -o r -- $CMD > $OUT/sq.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -d $OUT/mem -o r -- $CMD > $OUT/mem.log 2>&1
python - <<PY
import sqlite3, glob, collections
for sub in ("sq", "mem"):
    dbs = glob.glob("$OUT/%s/*results.db" % sub)
    if not dbs: print(sub, "no db"); continue
    c = sqlite3.connect(dbs[0]).cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    rows = c.execute("select kernel_name,counter_name,value from counters_collection order by start").fetchall()
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for kn, pn, v in rows:
        if "ray" in kn: acc[kn.split("(")[0][-40:]][pn].append(v)
    for kn, d in acc.items():
        print(sub, kn, {pn: round(sum(v[-6:]) / max(len(v[-6:]), 1)) for pn, v in d.items()})
PY
