#!/bin/bash
# Kernel time line of one step (start / end of consecutive kernels, gaps) at P = $1, and the kernel means.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
P=${1:-30}
OUT=$PWD/gpurun_out/r04_timeline_$P
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --memory-copy-trace --stats -d "$OUT/trace" -o p -- python bench.py --steps 8 --warmup 3 --no-cpu --sweep= --particles $P > "$OUT/trace.log" 2>&1
python tools/kernel_times.py "$OUT/trace/p_results.db" > "$OUT/kernel_times.txt" 2>&1
python - "$OUT/trace/p_results.db" > "$OUT/timeline.txt" 2>&1 <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end, grid_x from kernels order by start").fetchall()
try:
    cps = cur.execute("select name, start, end, size from memory_copies order by start").fetchall()
except Exception as e:
    cps = []
ev = [(s, e, n.split("(")[0].replace("void lama_dev::", "")[:44], g) for n, s, e, g in rows] + [(s, e, "COPY " + str(n)[:30], sz) for n, s, e, sz in cps]
ev.sort()
last = ev[-70:]
prev_end = None
for s, e, name, g in last:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{name:44s} {g:>9} dur {(e - s) / 1e3:9.2f} us  gap-to-prev-end {gap:9.2f} us")
    prev_end = max(e, prev_end or e)
PY
rm -rf "$OUT/trace"/*.db 2>/dev/null
tail -70 "$OUT/timeline.txt"
cat "$OUT/kernel_times.txt" | head -24
