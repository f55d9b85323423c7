#!/bin/bash
# in-situ A/B: ring depth (asm LDS-DMA) for the split-K 3x3 convs that run ONE workgroup per CU (8x8 / 4x4 levels)
B="python bench.py --no-cpu-baseline --no-psnr"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["config"]["ms_per_ddim_step"], d["value"])'
echo base $($B 2>/dev/null | python -c "$P")
for m in 512 2048; do for st in 3 4; do
  echo "M=$m stages=$st" $(NR_IGEMM_ADMA_MINK=16 NR_IGEMM_FORCE="-1,-1,-1,$st,-1,-1" NR_IGEMM_FORCE_MAXM=$m NR_IGEMM_FORCE_MINM=$m NR_IGEMM_FORCE_KS=3 $B 2>/dev/null | python -c "$P")
done; done
echo base $($B 2>/dev/null | python -c "$P")
