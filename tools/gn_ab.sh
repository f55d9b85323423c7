#!/bin/bash
# GroupNorm A/B: slab kernel (1024/512 threads) vs stats+apply passes; kernel durations from rocprofv3 (python launch overhead exceeds the kernels)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gn_slab -o gn -- python3 $R/tools/gn_bench.py > $R/gpurun_out/gn_slab.log 2>&1
export NR_GN_T=512
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gn_slab512 -o gn -- python3 $R/tools/gn_bench.py > $R/gpurun_out/gn_slab512.log 2>&1
unset NR_GN_T
export NR_GN_SLAB=0
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gn_old -o gn -- python3 $R/tools/gn_bench.py > $R/gpurun_out/gn_old.log 2>&1
cd $R
python3 tools/gn_ab_report.py
