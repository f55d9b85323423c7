# in-situ A/B of igemm overrides under hipGraph (no per-launch host overhead): bash tools/igemm_ab.sh
run() { python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-44s %.3f fps %.3f ms/step igemm %.0f TF/s' % ('$1', d['value'], d['config']['ms_per_ddim_step'], d['roofline']['achieved']))"; }
run base
export NR_IGEMM_FORCE_KS=3 NR_IGEMM_FORCE_MAXM=1024
NR_IGEMM_FORCE="-1,-1,-1,3,-1,-1" run "3x3 M<=1024 stages=3"
NR_IGEMM_FORCE="-1,-1,-1,4,-1,-1" run "3x3 M<=1024 stages=4"
export NR_IGEMM_FORCE_MAXM=2048 NR_IGEMM_FORCE_MINM=2048
NR_IGEMM_FORCE="-1,-1,-1,3,-1,-1" run "3x3 M=2048 stages=3"
NR_IGEMM_FORCE="-1,-1,-1,4,-1,-1" run "3x3 M=2048 stages=4"
unset NR_IGEMM_FORCE_KS NR_IGEMM_FORCE_MAXM NR_IGEMM_FORCE_MINM
run base2
