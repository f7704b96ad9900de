#!/bin/bash
# Profiling recipe of a round, run on the GPU box through gpurun:  bash tools/profile_round.sh <tag> [what ...]
# Writes under gpurun_out/prof_<tag>/; tools/summarize_rocprof.py turns the rocpd databases into the summaries committed under
# profiles/ (profiles/INDEX.md says which file backs which number of the bench line).  `what` (default: trace pmc):
#   trace     kernel trace + stats of the driver's bench command (exact brushfire, P = 30) and of the 3000-particle pool
#   pmc       FETCH_SIZE / WRITE_SIZE / SQ counters of the bench command, one counter group per run (PMC runs carry the kernel
#             trace only: no other tracing domain -- gpurun refuses the combination)
#   sq        two SQ counter groups of the brushfire kernel at 30 and 3000 particles (parked / issue split, branches)
#   timeline  start / end of consecutive kernels of the last steps at 3000 particles (gaps = host time between launches)
TAG=${1:-r05}; shift
WHAT=${*:-trace pmc}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
CMD="python bench.py --steps 20 --warmup 5 --no-cpu --sweep="      # the driver's command (--steps 20 --warmup 5)
for w in $WHAT; do
case $w in
trace)
  rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o $TAG -- $CMD > "$OUT/bench_trace.log" 2>&1
  rocprofv3 --kernel-trace --stats -d "$OUT/trace3000" -o $TAG -- $CMD --particles 3000 > "$OUT/bench_trace3000.log" 2>&1
  python tools/kernel_times.py "$OUT/trace3000/${TAG}_results.db" > "$OUT/kernel_times_3000.txt" 2>&1 ;;
pmc)
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o $TAG -- $CMD > "$OUT/bench_pmc_fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o $TAG -- $CMD > "$OUT/bench_pmc_write.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_BRANCH -d "$OUT/pmc_sq" -o $TAG -- $CMD > "$OUT/bench_pmc_sq.log" 2>&1 ;;
sq)
  A="SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_BUSY_CYCLES"
  B="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU"
  for P in 30 3000; do
    C8="python bench.py --steps 8 --warmup 3 --no-cpu --sweep= --particles $P"
    rocprofv3 --kernel-trace --pmc $A -d "$OUT/sqA_$P" -o p -- $C8 > "$OUT/sqA_$P.log" 2>&1
    rocprofv3 --kernel-trace --pmc $B -d "$OUT/sqB_$P" -o p -- $C8 > "$OUT/sqB_$P.log" 2>&1
    (python tools/pmc_kernel.py "$OUT/sqA_$P/p_results.db" "k_brushfire<1024"; python tools/pmc_kernel.py "$OUT/sqB_$P/p_results.db" "k_brushfire<1024") > "$OUT/sq_brushfire_$P.txt" 2>&1
    rm -rf "$OUT"/sq[AB]_$P
  done ;;
timeline)
  rocprofv3 --kernel-trace --stats -d "$OUT/tl" -o p -- python bench.py --steps 8 --warmup 3 --no-cpu --sweep= --particles 3000 > "$OUT/tl.log" 2>&1
  python - "$OUT/tl/p_results.db" > "$OUT/timeline_3000.txt" 2>&1 <<'PY'
import sqlite3, sys
rows = sqlite3.connect(sys.argv[1]).cursor().execute("select name, start, end, grid_x from kernels order by start").fetchall()
rows = [r for r in rows if "lama_dev" in r[0]][-90:]
t0, prev_end = rows[0][1], None
for name, s, e, g in rows:
    short = name.split("(")[0].replace("void lama_dev::", "")[:44]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{short:44s} grid {g:>8} start {(s - t0) / 1e3:10.2f} us dur {(e - s) / 1e3:9.2f} us  gap-to-latest-end {gap:9.2f} us")
    prev_end = max(e, prev_end or e)
PY
  rm -rf "$OUT/tl" ;;
esac
done
ls -la "$OUT"
