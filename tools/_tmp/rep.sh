cd "$GRAFT_REPO_ROOT"
for i in 1 2 3; do timeout 900 python -m pytest tests -m gpu -x -q -k "not hall_of_10000" 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -3; done
