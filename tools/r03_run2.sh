#!/bin/bash
# round-3 GPU session 2: fused FeedForward kernel (op test, standalone timing, in-situ A/B), a1 / C4 / C5 parity tests
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -m gpu -x -q -s -k "fused_feedforward" 2>&1 | tail -15 > gpurun_out/r03_ff_optest.log
python tools/ff_bench.py 32768 > gpurun_out/r03_ff_bench.log 2>&1
python tools/ff_bench.py 163840 >> gpurun_out/r03_ff_bench.log 2>&1
for v in 1 0; do
  NR_FF_FUSED=$v python bench.py --no-cpu-baseline --no-psnr --steps 3 --warmup 1 2>&1 | tail -1 > gpurun_out/r03_bench_ff$v.json
done
python -m pytest tests/test_a1_call.py tests/test_engine_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r03_t2_small.log
python -m pytest tests/test_c4c5_gpu.py -m gpu -x -q -s 2>&1 | tail -40 > gpurun_out/r03_t2_c4c5.log
tail -n 5 gpurun_out/r03_ff_optest.log gpurun_out/r03_ff_bench.log gpurun_out/r03_t2_small.log
