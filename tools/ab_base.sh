# same-box A/B of the working tree's library against neurons_amd/libneurons_amd_base.so (tools/ab_lib.sh <rev>; NR_LIB_VARIANT=base), interleaved:
# per-launch times of one U-Net forward, the headline, the keyframe loop, config 4.  Usage: bash tools/ab_base.sh out.txt [pattern for the per-op lines]
cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/ab_base.txt}
pat=${2:-lin160}
: > $out
for arm in base new; do
  v=""; [ $arm = base ] && v=base
  NR_LIB_VARIANT=$v python tools/per_op_profile.py gpurun_out/ab_${arm}_unet.csv gpurun_out/ab_${arm}_ctrl.csv > /dev/null 2>&1
  echo "--- per-op, $arm" >> $out
  python tools/per_op_buckets.py gpurun_out/ab_${arm}_unet.csv 400 | grep -E "ops,|$pat" >> $out
done
for rep in 1 2; do
  for arm in base new; do
    v=""; [ $arm = base ] && v=base
    NR_LIB_VARIANT=$v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-psnr --no-end-to-end 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline  $arm rep $rep:', d['value'], 'frames/s', d['config']['ms_per_ddim_step'], 'ms/DDIM step; class frac', d['roofline']['frac'])" >> $out
  done
done
for arm in base new; do
  v=""; [ $arm = base ] && v=base
  NR_LIB_VARIANT=$v python bench.py --workload keyframe --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('keyframe  $arm:', d['value'], 'keyframes/s', d['config']['ms_per_euler_step'], 'ms/Euler step; class frac', d['roofline']['frac'])" >> $out
done
for arm in base new; do
  v=""; [ $arm = base ] && v=base
  NR_LIB_VARIANT=$v python bench.py --workload video --batch 8 --steps 1 --warmup 1 --no-cpu-baseline --no-psnr --no-end-to-end 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('config 4 (8 clips)  $arm:', d['value'], 'frames/s; class frac', d['roofline']['frac'])" >> $out
done
cat $out
