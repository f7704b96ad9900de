#!/bin/bash
# Repeats tests/test_gpu_parity.py::test_config3_split_partition_invariance (G processes on ONE device against one context) and counts
# the runs that diverge: bash tools/flake_partition.sh <runs> <G> [<G> ...]   (on the GPU box)
cd "${GRAFT_REPO_ROOT:-.}"
export LAMA_TEST_EXTRA_WORLDS=5,10,12
N=${1:-20}; shift
for w in "$@"; do
  fails=0
  for i in $(seq 1 $N); do
    out=$(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "split_partition_invariance and $w]" 2>&1 | grep -E "passed|failed" | cut -c1-100 | tr '\n' ' ')
    case "$out" in *failed*) fails=$((fails+1));; esac
  done
  echo "G = $w processes: $fails divergent of $N runs"
done
