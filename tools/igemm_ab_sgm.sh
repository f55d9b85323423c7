# in-situ A/B on the sgm keyframe path (M = 512 weight-streaming GEMMs dominate): bash tools/igemm_ab_sgm.sh
run() { python bench.py --workload keyframe --steps 1 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-44s %.4f kf/s %.3f ms/euler-step' % ('$1', d['value'], d['config']['ms_per_euler_step']))"; }
run base
export NR_IGEMM_FORCE_KS=1 NR_IGEMM_FORCE_MAXM=512
NR_IGEMM_FORCE="-1,-1,1,-1,-1,-1" run "1x1 M<=512 no split-K"
NR_IGEMM_FORCE="-1,-1,2,-1,-1,-1" run "1x1 M<=512 split-K 2 everywhere"
export NR_IGEMM_FORCE_MAXM=2048 NR_IGEMM_FORCE_MINM=1024
NR_IGEMM_FORCE="-1,-1,-1,4,-1,-1" run "1x1 M=2048 stages=4 (plan tile)"
NR_IGEMM_FORCE="64,64,-1,4,-1,4" run "1x1 M=2048 64x64 stages=4"
NR_IGEMM_FORCE="64,32,-1,4,-1,4" run "1x1 M=2048 64x32 stages=4"
unset NR_IGEMM_FORCE_KS NR_IGEMM_FORCE_MAXM NR_IGEMM_FORCE_MINM
run base2
