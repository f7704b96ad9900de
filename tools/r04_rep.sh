#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu --sweep= > /tmp/b.json 2>/tmp/b.err
  python - <<'PY'
import json
try:
    j = json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(round(j["value"]), round(j["ms_per_step"], 4), round(j["kernel_ms_per_step"]["brushfire"], 4))
except Exception as e:
    print("fail", e, open('/tmp/b.err').read()[-300:])
PY
done
