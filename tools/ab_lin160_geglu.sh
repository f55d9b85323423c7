# same-box A/B of the LayerNorm-folded GEGLU variant of lin160.hip (NR_LIN160_GEGLU=0: tiled igemm) on the keyframe path and the headline, interleaved
cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/lin160_geglu_ab.txt}
: > $out
for rep in 1 2; do
  for arm in 0 1; do
    NR_LIN160_GEGLU=$arm python bench.py --workload keyframe --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('keyframe  NR_LIN160_GEGLU=$arm rep $rep:', d['value'], 'keyframes/s', d['config']['ms_per_euler_step'], 'ms/Euler step; class frac', d['roofline']['frac'])" >> $out
  done
done
for arm in 0 1; do
  NR_LIN160_GEGLU=$arm python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-psnr --no-end-to-end 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline  NR_LIN160_GEGLU=$arm:', d['value'], 'frames/s', d['config']['ms_per_ddim_step'], 'ms/DDIM step')" >> $out
done
cat $out
