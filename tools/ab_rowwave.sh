# same-box A/B of the row-wave 128x160 tile rule (NR_IGEMM_ROWWAVE) on the keyframe and headline workloads
for i in 1 2; do
  for v in 0 1; do
    NR_IGEMM_ROWWAVE=$v python bench.py --workload keyframe --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('keyframe NR_IGEMM_ROWWAVE=$v', d['value'], d['ms_per_step'])"
  done
done
for i in 1 2 3; do
  for v in 0 1; do
    NR_IGEMM_ROWWAVE=$v python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline NR_IGEMM_ROWWAVE=$v', d['value'], d['ms_per_step'])"
  done
done
