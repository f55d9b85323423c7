cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do
  timeout 300 python -m pytest tests/test_engine_gpu.py -q -k "batch_of_clips or overlapped or prefetched or context_cache" --tb=line 2>&1 | grep -E "passed|failed|Error|rror:" | head -4
done
