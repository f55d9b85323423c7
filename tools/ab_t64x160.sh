# same-box A/B of the 64x160 one-round tile rule (NR_IGEMM_T64X160) on the headline and keyframe workloads
for i in 1 2 3; do
  for v in 0 1; do
    NR_IGEMM_T64X160=$v python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline NR_IGEMM_T64X160=$v', d['value'], d['ms_per_step'], d['config'].get('psnr_c2_vs_fp32_oracle_db'))"
  done
done
for v in 0 1; do
  NR_IGEMM_T64X160=$v python bench.py --workload keyframe --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('keyframe NR_IGEMM_T64X160=$v', d['value'], d['ms_per_step'])"
done
