#!/bin/bash
# Round evidence on ONE box: default bench (with CPU baseline), kernel stats, HBM traffic, MFMA utilisation, per-op CSVs,
# keyframe / VAE / config-4 / config-5 lines.  Outputs under gpurun_out/ev/ ; copy the summaries to profiles/.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/ev
cd $R
python bench.py > gpurun_out/ev/bench.log 2>&1
python bench.py --workload keyframe > gpurun_out/ev/bench_keyframe.log 2>&1
python bench.py --workload keyframe --keyframe-latent 96 --keyframe-steps 38 --no-cpu-baseline > gpurun_out/ev/bench_keyframe96.log 2>&1
python bench.py --workload keyframe --batch 8 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/ev/bench_keyframe_b8.log 2>&1
python bench.py --workload enhance --batch 8 --steps 1 --warmup 1 > gpurun_out/ev/bench_enhance_b8.log 2>&1
python bench.py --workload vae > gpurun_out/ev/bench_vae.log 2>&1
python bench.py --batch 8 --steps 1 --warmup 1 --no-cpu-baseline --no-psnr > gpurun_out/ev/bench_c4_batch8.log 2>&1
python bench.py --frames 32 --latent 64 --batch 4 --steps 1 --warmup 1 > gpurun_out/ev/bench_c5_bf16.log 2>&1
python bench.py --frames 32 --latent 64 --batch 4 --steps 1 --warmup 1 --attn-fp8 > gpurun_out/ev/bench_c5_fp8.log 2>&1
python tools/per_op_profile.py gpurun_out/ev/per_op_unet.csv gpurun_out/ev/per_op_ctrl.csv 1 16 32 5 > gpurun_out/ev/per_op.log 2>&1
python tools/per_op_profile.py gpurun_out/ev/per_op_c4_unet.csv gpurun_out/ev/per_op_c4_ctrl.csv 8 16 32 4 > gpurun_out/ev/per_op_c4.log 2>&1
python tools/launch_floor.py > gpurun_out/ev/launch_floor.txt 2>&1
bash tools/profile_round.sh > gpurun_out/ev/profile_round.log 2>&1
cp gpurun_out/traffic_pmc.json gpurun_out/ev/traffic_pmc.json
cp gpurun_out/prof_stats/stats_kernel_stats.csv gpurun_out/ev/rocprofv3_kernel_stats.csv
bash tools/profile_mfma.sh > gpurun_out/ev/profile_mfma.log 2>&1
cp gpurun_out/mfma_util.json gpurun_out/ev/mfma_util_pmc.json
bash tools/profile_keyframe.sh > gpurun_out/ev/profile_keyframe.log 2>&1
cp gpurun_out/kf_traffic_pmc.json gpurun_out/ev/keyframe_traffic_pmc.json
cp gpurun_out/kf_stats/stats_kernel_stats.csv gpurun_out/ev/keyframe_rocprofv3_kernel_stats.csv
tail -c 300 gpurun_out/ev/bench.log
