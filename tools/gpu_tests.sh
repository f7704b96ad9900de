#!/bin/bash
# the -m gpu suite on the GPU box; log under gpurun_out/
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -m gpu -x -q ${1:+-k "$1"} > gpurun_out/gpu_tests.log 2>&1
echo "rc=$?" >> gpurun_out/gpu_tests.log
grep -E "passed|failed|error|rc=" gpurun_out/gpu_tests.log | tail -5
