#!/bin/bash
# Developer tool: SQ / memory counters of the ray-cast kernels at a given particle count (two PMC passes, kernel trace only).
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
P=${1:-3000}
OUT=gpurun_out/pmc_ray_$P; mkdir -p $OUT
CMD="python bench.py --no-cpu --particles $P --steps 6 --warmup 3 --sweep="
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES -d $OUT/sq -o r -- $CMD > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE TCC_EA_ATOMIC_sum TCC_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/mem -o r -- $CMD > $OUT/mem.log 2>&1
python - <<PY
import sqlite3, glob, collections
for sub in ("sq", "mem"):
    dbs = glob.glob("$OUT/%s/*results.db" % sub)
    if not dbs: print(sub, "no db"); continue
    c = sqlite3.connect(dbs[0]).cursor()
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    pmc = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    rows = c.execute("""select k.kernel_name, p.name, e.value from %s e
        join rocpd_info_pmc p on p.id = e.pmc_id
        join rocpd_kernel_dispatch d on d.event_id = e.event_id
        join rocpd_info_kernel_symbol k on k.id = d.kernel_id""" % pmc).fetchall()
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for kn, pn, v in rows:
        if "ray" in kn: acc[kn.split("(")[0][-40:]][pn].append(v)
    for kn, d in acc.items():
        print(sub, kn, {pn: round(sum(v[-6:]) / max(len(v[-6:]), 1)) for pn, v in d.items()})
PY
