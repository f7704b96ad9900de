#!/bin/bash
# one-device exercise of the multi-GPU paths: sharding tests + `bench.py --gpus 2` with every rank on GPU 0 (gloo)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$PWD/gpurun_out/r04_multi; mkdir -p "$OUT"
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3 or multi_gpu_object or sharded" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"
grep -E "passed|failed|rc=|skipped" "$OUT/pytest.log" | tail -3
LAMA_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus ${1:-2} --steps 6 --warmup 2 --no-cpu > "$OUT/bench_multi.json" 2> "$OUT/bench_multi.err"; echo "bench rc=$?"
python - "$OUT/bench_multi.json" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    for k in ("value", "ms_per_step", "n_gpus", "scaling", "value_source", "torch_distributed_ranks", "strong_scaling_ceiling", "weak_scaling", "single_gpu_same_pool"):
        print(k, j.get(k))
except Exception as e:
    print("parse failed", e); print(open(sys.argv[1]).read()[-1500:])
PY
tail -5 "$OUT/bench_multi.err"
