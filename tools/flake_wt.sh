for d in _wt/r1 _wt/c2; do
  cd $GRAFT_REPO_ROOT/$d
  for i in 1 2 3 4 5; do
    r=$(timeout 300 python -m pytest tests/test_engine_gpu.py -q -k "batch_of_clips or overlapped or prefetched or context_cache" --tb=line 2>&1 | grep -E "passed|failed" | tail -1)
    echo "$d run $i: $r"
  done
done
