cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for P in 300 3000; do
LAMA_HIP_SEQUENTIAL_RAYCAST=2 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_ray_$P -o r -- python bench.py --no-cpu --particles $P --steps 10 --warmup 3 --sweep "" > gpurun_out/prof_ray_$P.log 2>&1
python - <<PY
import sqlite3,glob
db=glob.glob("gpurun_out/prof_ray_$P/*results.db")[0]
for name,calls,tot,avg,pct in sqlite3.connect(db).cursor().execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if "lama_dev" in name: print($P, name.split('(')[0][:60], calls, round(avg/1e3,1),"us")
PY
done
