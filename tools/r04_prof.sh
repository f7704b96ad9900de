#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$PWD/gpurun_out/r04_prof; mkdir -p "$OUT"
P=${1:-30}
for v in ${2:-prof main count}; do
  case $v in
    prof) LAMA_PROF_LIB=tools/_prof/liblama_hip_prof.so python tools/prof_bf.py $P > "$OUT/prof_$P.txt" 2>&1; tail -4 "$OUT/prof_$P.txt";;
    main) LAMA_PROF_MAIN=1 LAMA_PROF_LIB=tools/_prof/liblama_hip_prof_main.so python tools/prof_bf.py $P > "$OUT/prof_main_$P.txt" 2>&1; tail -4 "$OUT/prof_main_$P.txt";;
    fine) LAMA_PROF_MAIN=1 LAMA_PROF_FINE=1 LAMA_PROF_LIB=tools/_prof/liblama_hip_prof_fine.so python tools/prof_bf.py $P > "$OUT/prof_fine_$P.txt" 2>&1; tail -4 "$OUT/prof_fine_$P.txt";;
    count) LAMA_PROF_MAIN=1 LAMA_PROF_COUNT=1 LAMA_PROF_LIB=tools/_prof/liblama_hip_prof_count.so python tools/prof_bf.py $P > "$OUT/prof_count_$P.txt" 2>&1; tail -4 "$OUT/prof_count_$P.txt";;
  esac
done
