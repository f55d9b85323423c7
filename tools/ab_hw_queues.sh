# Do the two engine streams (U-Net, SparseCtrl) really run concurrently?  GPU_MAX_HW_QUEUES caps the hardware queues HIP streams are mapped onto
# (default 4): with 1 every stream shares one queue (kernels of the two networks serialise).  Same box, interleaved.
cd $GRAFT_REPO_ROOT
out=${1:-gpurun_out/hw_queues_ab.txt}
: > $out
for rep in 1 2; do
  for q in 1 default 8; do
    if [ "$q" = "default" ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-psnr --no-end-to-end 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('GPU_MAX_HW_QUEUES=$q rep $rep:', d['value'], 'frames/s', d['config']['ms_per_ddim_step'], 'ms/DDIM step')" >> $out
  done
done
unset GPU_MAX_HW_QUEUES
cat $out
