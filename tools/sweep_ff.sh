#!/bin/bash
export SWEEP_SET=ff
python tools/gemm_sweep.py > gpurun_out/r02_sweep_ff.txt 2>&1
NR_ROWPANEL=0 python tools/gemm_sweep.py > gpurun_out/r02_sweep_ff_norp.txt 2>&1
