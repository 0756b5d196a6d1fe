R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for route in quads tokens; do echo "== $route"; CASMTR_CALLER_LAYOUT=$route timeout 600 python -m pytest tests/test_model_harness.py -x -q -m gpu -k coarse_stage 2>&1 | grep -E "AssertionError|passed|failed|within" | head -5; done
