#!/bin/bash
# Round-4 evidence run (GPU box, through gpurun): the single-wave issue-cost table, the SQ counters VERDICT r03 asked for
# (parked / issue split, branches, scalar memory) of the brushfire at 30 and 3000 particles, and the kernel time line of the
# 3000-particle step (start / end of consecutive kernels: does the resume stage overlap the first stage's tail?).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_probe
rm -rf "$OUT"; mkdir -p "$OUT"
tools/_prof/issue_costs > "$OUT/issue_costs.txt" 2>&1
rocprofv3 -L > "$OUT/counters_list.txt" 2>&1
A="SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INST_CYCLES_SALU"
B="SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_INSTS_CBRANCH"
for P in 30 3000; do
  CMD="python bench.py --steps 8 --warmup 3 --no-cpu --sweep= --particles $P"
  rocprofv3 --kernel-trace --pmc $A -d "$OUT/pmcA_$P" -o p -- $CMD > "$OUT/pmcA_$P.log" 2>&1
  rocprofv3 --kernel-trace --pmc $B -d "$OUT/pmcB_$P" -o p -- $CMD > "$OUT/pmcB_$P.log" 2>&1
  python tools/pmc_kernel.py "$OUT/pmcA_$P/p_results.db" k_brushfire > "$OUT/sq_brushfire_$P.txt" 2>&1
  python tools/pmc_kernel.py "$OUT/pmcB_$P/p_results.db" k_brushfire >> "$OUT/sq_brushfire_$P.txt" 2>&1
done
rocprofv3 --kernel-trace --stats -d "$OUT/trace3000" -o p -- python bench.py --steps 8 --warmup 3 --no-cpu --sweep= --particles 3000 > "$OUT/trace3000.log" 2>&1
python tools/kernel_times.py "$OUT/trace3000/p_results.db" > "$OUT/kernel_times_3000.txt" 2>&1
python - "$OUT/trace3000/p_results.db" > "$OUT/timeline_3000.txt" 2>&1 <<'EOF'
import sqlite3, sys
rows = sqlite3.connect(sys.argv[1]).cursor().execute("select name, start, end, grid_x from kernels order by start").fetchall()
rows = [r for r in rows if "lama_dev" in r[0]]
# the last 12 map updates: every kernel with its start relative to the previous kernel's end (negative = overlap)
last = rows[-180:]
prev_end = None
for name, s, e, gx in last:
    short = name.split("(")[0].replace("void lama_dev::", "")[:44]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{short:44s} grid {gx:>8} dur {(e - s) / 1e3:9.2f} us  gap-to-prev-end {gap:9.2f} us")
    prev_end = e
EOF
ls -la "$OUT"
cat "$OUT/issue_costs.txt"
