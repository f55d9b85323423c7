# same-box A/B of the register-panel form of lin160.hip (NR_LIN160_PANEL_CGMAJOR=0: LayerNorm launch / folded tiled igemm as before) on the headline, interleaved (ABAB),
# + per-launch times of one U-Net forward in both arms (tools/per_op_profile.py), + config 4
cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/lin160_panel_cgmajor_ab.txt}
: > $out
for arm in 0 1; do
  NR_LIN160_PANEL_CGMAJOR=$arm python tools/per_op_profile.py gpurun_out/pcg_${arm}_unet.csv gpurun_out/pcg_${arm}_ctrl.csv > /dev/null 2>&1
  echo "--- per-op, NR_LIN160_PANEL_CGMAJOR=$arm" >> $out
  python tools/per_op_buckets.py gpurun_out/pcg_${arm}_unet.csv 400 | grep -E "ops,|geglu|lin160|layernorm|N=1920|N=3840" >> $out
done
for rep in 1 2; do
  for arm in 0 1; do
    NR_LIN160_PANEL_CGMAJOR=$arm python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-psnr --no-end-to-end 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline  NR_LIN160_PANEL_CGMAJOR=$arm rep $rep:', d['value'], 'frames/s', d['config']['ms_per_ddim_step'], 'ms/DDIM step; class frac', d['roofline']['frac'])" >> $out
  done
done
for arm in 0 1; do
  NR_LIN160_PANEL_CGMAJOR=$arm python bench.py --workload video --batch 8 --steps 1 --warmup 1 --no-cpu-baseline --no-psnr --no-end-to-end 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config 4 (8 clips)  NR_LIN160_PANEL_CGMAJOR=$arm:', d['value'], 'frames/s; class frac', d['roofline']['frac'])" >> $out
done
cat $out
