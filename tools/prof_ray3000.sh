#!/bin/bash
# Developer recipe: per-kernel times and SQ counters of the chip-full regime (3000 particles), ray-cast kernels in particular.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_ray3000
rm -rf "$OUT"; mkdir -p "$OUT"
CMD="python bench.py --steps 8 --warmup 2 --no-cpu --sweep= --particles 3000"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o ray -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT -d "$OUT/pmc_sq" -o ray -- $CMD > "$OUT/pmc.log" 2>&1
python tools/kernel_times.py "$OUT/trace/ray_results.db" > "$OUT/kernel_times.txt" 2>&1
python tools/pmc_kernel.py "$OUT/pmc_sq/ray_results.db" k_ray > "$OUT/pmc_ray.txt" 2>&1
grep "3000x" "$OUT/kernel_times.txt" | head -20
grep -E " 768000| 3000 |grid +[0-9]+ " "$OUT/pmc_ray.txt" | head -60
