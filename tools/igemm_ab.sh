# in-situ A/B of igemm overrides under hipGraph (no per-launch host overhead): bash tools/igemm_ab.sh
run() { python bench.py --no-cpu-baseline --no-psnr --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-44s %.3f fps %.3f ms/step igemm %.0f TF/s' % ('$1', d['value'], d['config']['ms_per_ddim_step'], d['roofline']['achieved']))"; }
run base
export NR_IGEMM_FORCE_KS=3 NR_IGEMM_FORCE_MAXM=8192 NR_IGEMM_FORCE_MINM=8192
NR_IGEMM_FORCE="128,160,1,2,-1,4" run "3x3 M=8192 128x160 split1"
NR_IGEMM_FORCE="128,160,2,2,-1,4" run "3x3 M=8192 128x160 split2"
NR_IGEMM_FORCE="128,160,4,2,-1,4" run "3x3 M=8192 128x160 split4"
NR_IGEMM_FORCE="128,128,2,2,-1,8" run "3x3 M=8192 128x128/8w split2"
export NR_IGEMM_FORCE_MAXM=32768 NR_IGEMM_FORCE_MINM=32768
NR_IGEMM_FORCE="128,160,1,2,-1,4" run "3x3 M=32768 128x160 split1"
NR_IGEMM_FORCE="128,160,2,2,-1,4" run "3x3 M=32768 128x160 split2"
unset NR_IGEMM_FORCE_KS NR_IGEMM_FORCE_MAXM NR_IGEMM_FORCE_MINM
run base2
