R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/trace1 -o tr --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --ddim-steps 8 --no-cpu-baseline --no-psnr > $R/gpurun_out/trace1.log 2>&1
cd $R
ls -la gpurun_out/trace1/*/ 2>/dev/null | head; find gpurun_out/trace1 -name "*kernel_trace.csv" | head -2
