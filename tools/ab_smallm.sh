python -m pytest tests/test_sgm_gpu.py tests/test_leaf_gpu.py tests/test_fullsize_gpu.py tests/test_ops_gpu.py -x -q 2>&1 | tail -5
for i in 1 2; do
  for v in 0 1; do
    echo "== keyframe NR_SMALLM=$v"; NR_SMALLM=$v python bench.py --workload keyframe --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
for i in 1 2; do
  for v in 0 1; do
    echo "== headline NR_SMALLM=$v"; NR_SMALLM=$v python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  done
done
