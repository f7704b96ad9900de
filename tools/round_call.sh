#!/bin/bash
# One gpurun call of round 6: the -m gpu suite, then the driver's bench command; logs under gpurun_out/r06_<tag>/
# usage: bash tools/round_call.sh <tag> [sweep scale (0 = no determinism sweep)]
TAG=${1:-a}
SWEEP=${2:-0}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r06_$TAG
mkdir -p "$OUT"
timeout 2400 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests.log" 2>&1
echo "rc=$?" >> "$OUT/gpu_tests.log"
tail -4 "$OUT/gpu_tests.log"
timeout 900 python bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"
python - <<PY
import json
try:
    d = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms/step", d["ms_per_step"], "kernels", d["kernel_ms_per_step"])
    print("memory", d.get("memory"))
    for k, v in d.get("other_particle_counts", {}).items():
        print(k, v["value"], v["ms_per_step"], v["brushfire_ms"], v["raycast_ms"], v["scan_match_ms"], v.get("memory"))
    print("resample_3000", d.get("resample_3000"))
    print("forced", d.get("forced_resample_variant"))
    nr = d.get("next_rows", {})
    print("slam2d", nr.get("slam2d_update_cfg4"), "loc2d", nr.get("loc2d_update_cfg1"))
    print("map_load", nr.get("loc2d_map_load"), nr.get("error_loc2d_map_load"))
    print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "cores", "kind")})
except Exception as e:
    print("parse error", e)
    print(open("$OUT/bench.err").read()[-3000:])
PY
if [ "$SWEEP" != "0" ]; then
  timeout 2400 python tools/determinism_sweep.py $SWEEP > "$OUT/determinism_sweep.txt" 2>&1
  echo "sweep rc=$?"; grep -v "amdgpu.ids" "$OUT/determinism_sweep.txt" | tail -12
fi
