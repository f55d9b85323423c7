# same-box A/B: ring depth of the split-K 128x160 tile on the 3x3 convs with M <= 512 (HBM-cold 29.5 MB weight tensors)
for i in 1 2; do
  for st in 2 3 4; do
    NR_IGEMM_FORCE="128,160,-1,$st,-1,4" NR_IGEMM_FORCE_KS=3 NR_IGEMM_FORCE_MAXM=512 python bench.py --workload keyframe --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('keyframe conv ring $st', d['value'], d['ms_per_step'])"
  done
done
for i in 1 2; do
  for st in 2 3 4; do
    NR_IGEMM_FORCE="128,160,-1,$st,-1,4" NR_IGEMM_FORCE_KS=3 NR_IGEMM_FORCE_MAXM=512 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline conv ring $st', d['value'], d['ms_per_step'])"
  done
done
