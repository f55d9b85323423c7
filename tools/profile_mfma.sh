R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_mfma -o mfma --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --ddim-steps 4 --no-cpu-baseline --no-psnr --no-op-profile > $R/gpurun_out/pmc_mfma.log 2>&1
cd $R
python3 tools/pmc_mfma.py gpurun_out/pmc_mfma gpurun_out/mfma_util.json
find gpurun_out/pmc_mfma -name "*.csv" -size +2M -delete
