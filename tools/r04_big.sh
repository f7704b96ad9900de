#!/bin/bash
# pace of the resume-stage instantiation: P = 30 with every particle handed over after its first pop
cd "$GRAFT_REPO_ROOT" || exit 1
for b in "" "1,30"; do
  LAMA_HIP_BF_BUDGET=$b python bench.py --steps 20 --warmup 5 --no-cpu --sweep= > /tmp/b.json 2>/tmp/b.err
  [ -z "$b" ] && unset LAMA_HIP_BF_BUDGET
  python - "$b" <<'PY'
import json, sys
try:
    j = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print("budget", repr(sys.argv[1]), round(j["value"]), round(j["ms_per_step"], 4), round(j["kernel_ms_per_step"]["brushfire"], 4), j["config"]["kernels_that_ran"])
except Exception as e:
    print("fail", e, open('/tmp/b.err').read()[-300:])
PY
done
