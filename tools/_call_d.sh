cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05_d
timeout 1500 python -m pytest tests -m gpu -x -q -s -k "hall_of_10000" > gpurun_out/r05_d/sel.log 2>&1; echo "rc=$?" >> gpurun_out/r05_d/sel.log; grep -E "big world|passed|failed|rc=|Error" gpurun_out/r05_d/sel.log | tail
bash tools/profile_round.sh r05 trace timeline > gpurun_out/r05_d/prof.log 2>&1
tail -5 gpurun_out/r05_d/prof.log
