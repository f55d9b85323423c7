cd $GRAFT_REPO_ROOT
for mk in 100000 64 24 12 4 0; do
  NR_IGEMM_ADMA_MINK=$mk python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-psnr 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ADMA_MINK=$mk', d['value'], 'fps', d['config']['ms_per_ddim_step'], 'ms/step igemm', d['roofline']['ms_per_ddim_step'])"
done
