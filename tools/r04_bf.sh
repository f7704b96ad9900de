#!/bin/bash
# Round-4 developer loop (GPU box): brushfire parity subset, then the bench at 30 and 3000 particles.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r04_bf
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "${1:-stagewise or round_room or randomized_rooms_maps or free_running_host or slam2d_online or lidar}" > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -5 "$OUT/pytest.log"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu --sweep=3000 > "$OUT/bench.json" 2> "$OUT/bench.err"
python - "$OUT/bench.json" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("value", j["value"], "ms/step", j["ms_per_step"], "kernels", {k: round(v, 4) for k, v in j["kernel_ms_per_step"].items() if isinstance(v, float)})
    o = j.get("other_particle_counts", {})
    for k, v in o.items():
        print(k, {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items() if kk in ("value", "ms_per_step", "kernel_ms_per_step")})
    print("next_rows", {k: v.get("gpu_ms_per_update") for k, v in j.get("next_rows", {}).items() if isinstance(v, dict) and "gpu_ms_per_update" in v})
except Exception as e:
    print("bench parse failed", e); print(open(sys.argv[1]).read()[-2000:])
PY
tail -3 "$OUT/bench.err"
