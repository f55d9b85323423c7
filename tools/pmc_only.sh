#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch -o fetch --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --ddim-steps 8 --no-cpu-baseline --no-psnr --no-op-profile > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write -o write --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --ddim-steps 8 --no-cpu-baseline --no-psnr --no-op-profile > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 8 gpurun_out/traffic_pmc.json > /dev/null
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" -size +2M -delete
