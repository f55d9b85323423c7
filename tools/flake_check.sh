# repeat the call-path equivalence tests (separate vs overlapped vs prefetched vs batched) to catch timing-dependent bugs
for mode in default 1; do
  for i in 1 2 3; do
    if [ "$mode" = default ]; then unset NR_LN_FUSE; else export NR_LN_FUSE=$mode; fi
    r=$(timeout 300 python -m pytest tests/test_engine_gpu.py -x -q -k "batch_of_clips or overlapped or prefetched or context_cache" 2>&1 | grep -E "passed|failed" | tail -1)
    echo "LN_FUSE=$mode run $i: $r"
  done
done
