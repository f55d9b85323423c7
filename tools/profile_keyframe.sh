#!/bin/bash
# BASELINE config 3 evidence: kernel stats + HBM traffic (separate PMC passes) of the sgm unCLIP Euler loop
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kf_stats -o stats --output-format csv -- python3 $R/bench.py --workload keyframe --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/kf_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/kf_fetch -o fetch --output-format csv -- python3 $R/bench.py --workload keyframe --steps 1 --warmup 0 --keyframe-steps 4 --no-cpu-baseline > $R/gpurun_out/kf_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/kf_write -o write --output-format csv -- python3 $R/bench.py --workload keyframe --steps 1 --warmup 0 --keyframe-steps 4 --no-cpu-baseline > $R/gpurun_out/kf_write.log 2>&1
cd $R
python3 tools/pmc_traffic.py gpurun_out/kf_fetch gpurun_out/kf_write 5 gpurun_out/kf_traffic_pmc.json
find gpurun_out/kf_fetch gpurun_out/kf_write gpurun_out/kf_stats -name "*.csv" ! -name "*kernel_stats.csv" -size +2M -delete
