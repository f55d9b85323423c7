# same-box A/B of the round-6 short-K Linear kernel (lin160.hip): NR_LIN160=0 restores the tiled igemm.  Interleaved (ABAB).
cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/lin160_ab.txt}
: > $out
for rep in 1 2; do
  for arm in 0 1; do
    NR_LIN160=$arm python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-psnr --no-end-to-end 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline  NR_LIN160=$arm rep $rep:', d['value'], 'frames/s', d['config']['ms_per_ddim_step'], 'ms/DDIM step; class frac', d['roofline']['frac'])" >> $out
  done
done
for arm in 0 1; do
  NR_LIN160=$arm python bench.py --workload video --batch 8 --steps 1 --warmup 1 --no-cpu-baseline --no-psnr --no-end-to-end 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config 4 (8 clips)  NR_LIN160=$arm:', d['value'], 'frames/s; class frac', d['roofline']['frac'])" >> $out
done
cat $out
