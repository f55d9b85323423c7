#!/bin/bash
# pipeline-depth sweep with the asm LDS-DMA (tiles really in flight): default ADMA threshold, then forced on for every K
export SWEEP_CONFIGS="128,128,1,2,-1,8;128,128,1,3,-1,8;128,128,1,4,-1,8;128,160,1,2,-1,4;128,160,1,3,-1,4;128,160,1,4,-1,4;256,128,1,2,-1,8;256,128,1,3,-1,8;128,64,1,3,-1,4;128,64,1,4,-1,4"
python tools/gemm_sweep.py > gpurun_out/r02_sweep_stages_default.txt 2>&1
NR_IGEMM_ADMA_MINK=0 python tools/gemm_sweep.py > gpurun_out/r02_sweep_stages_adma_all.txt 2>&1
NR_IGEMM_ADMA_MINK=100000 python tools/gemm_sweep.py > gpurun_out/r02_sweep_stages_builtin_all.txt 2>&1
