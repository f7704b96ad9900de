#!/bin/bash
# Profiling recipe of a round (tag = first argument, default r02) (run on the GPU box through gpurun).  Writes under gpurun_out/; tools/summarize_rocprof.py
# turns the rocpd databases into the committed summaries under profiles/.
#   pass 1: kernel trace + stats of the default bench command (exact brushfire, P = 30)
#   pass 2-4: PMC counters in separate runs (no tracing domains besides the kernel trace)
TAG=${1:-r04}
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
STEPS=${2:-20}
CMD="python bench.py --steps $STEPS --warmup 5 --no-cpu --sweep="      # the driver's command (--steps 20 --warmup 5)
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o $TAG -- $CMD > "$OUT/bench_trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o $TAG -- $CMD > "$OUT/bench_pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/pmc_write" -o $TAG -- $CMD > "$OUT/bench_pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_BRANCH -d "$OUT/pmc_sq" -o $TAG -- $CMD > "$OUT/bench_pmc_sq.log" 2>&1
# kernel trace of the 3000-particle pool on one GPU (per-kernel times of the chip-full regime)
rocprofv3 --kernel-trace --stats -d "$OUT/trace3000" -o $TAG -- python bench.py --steps $STEPS --warmup 5 --no-cpu --sweep= --particles 3000 > "$OUT/bench_trace3000.log" 2>&1
# (the summaries are written by tools/summarize_rocprof.py in the build container, from the databases this call brings back)
python tools/kernel_times.py "$OUT/trace3000/${TAG}_results.db" > "$OUT/kernel_times_3000.txt" 2>&1
ls -la "$OUT"/*
