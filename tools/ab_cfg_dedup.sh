# same-box A/B of the exact CFG de-duplication in the U-Net (NR_CFG_DEDUP=0 restores the full evaluation), interleaved.  Usage: bash tools/ab_cfg_dedup.sh [out.txt]
cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/cfg_dedup_ab.txt}
: > $out
for rep in 1 2; do
  for arm in 0 1; do
    NR_CFG_DEDUP=$arm python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-psnr --no-end-to-end 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline  NR_CFG_DEDUP=$arm rep $rep:', d['value'], 'frames/s', d['config']['ms_per_ddim_step'], 'ms/DDIM step; class frac', d['roofline']['frac'], 'psnr', d['config'].get('psnr_c2_vs_fp32_oracle_db'))" >> $out
  done
done
cat $out
