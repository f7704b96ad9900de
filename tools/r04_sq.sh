#!/bin/bash
# SQ counters of the brushfire kernel at 30 and 3000 particles (two passes of 8 counters each)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_sq
rm -rf "$OUT"; mkdir -p "$OUT"
A="SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_BUSY_CYCLES"
B="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU"
for P in ${1:-30 3000}; do
  CMD="python bench.py --steps 8 --warmup 3 --no-cpu --sweep= --particles $P"
  rocprofv3 --kernel-trace --pmc $A -d "$OUT/pmcA_$P" -o p -- $CMD > "$OUT/pmcA_$P.log" 2>&1
  rocprofv3 --kernel-trace --pmc $B -d "$OUT/pmcB_$P" -o p -- $CMD > "$OUT/pmcB_$P.log" 2>&1
  (python tools/pmc_kernel.py "$OUT/pmcA_$P/p_results.db" "k_brushfire<1024"; python tools/pmc_kernel.py "$OUT/pmcB_$P/p_results.db" "k_brushfire<1024") | grep -v "grid      128" > "$OUT/sq_brushfire_$P.txt" 2>&1
  python tools/kernel_times.py "$OUT/pmcA_$P/p_results.db" | head -12 > "$OUT/kernel_times_$P.txt"
  cat "$OUT/sq_brushfire_$P.txt"; head -4 "$OUT/kernel_times_$P.txt"
  rm -rf "$OUT"/pmc*_$P
done
