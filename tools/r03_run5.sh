#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -s -k "fused_" 2>&1 | tail -12 > gpurun_out/r03_v2_optest.log
python tools/ff_bench.py 32768 > gpurun_out/r03_v2_bench.log 2>&1
python tools/tattn_bench.py >> gpurun_out/r03_v2_bench.log 2>&1
bash tools/fused_pmc.sh r03v2 > /dev/null 2>&1
grep -h "ff_fused\|tattn_fused" gpurun_out/r03v2_pmc_summary.txt | sort -u > gpurun_out/r03v2_pmc_short.txt
for v in 1 0; do
  NR_FF_FUSED=$v NR_TATTN_FUSED=$v python bench.py --no-cpu-baseline --no-psnr --steps 3 --warmup 1 2>&1 | tail -1 > gpurun_out/r03_v2_bench_fused$v.json
done
tail -n 8 gpurun_out/r03_v2_optest.log gpurun_out/r03_v2_bench.log gpurun_out/r03v2_pmc_short.txt
