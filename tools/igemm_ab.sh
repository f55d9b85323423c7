run() { python bench.py --no-cpu-baseline --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-44s %.3f fps %.3f ms/step' % ('$1', d['value'], d['config']['ms_per_ddim_step']))"; }
run base
export NR_IGEMM_FORCE_KS=1 NR_IGEMM_FORCE_MAXM=2048
NR_IGEMM_FORCE="-1,-1,-1,3,-1,-1" run "1x1 M<=2048 stages=3"
NR_IGEMM_FORCE="-1,-1,-1,4,-1,-1" run "1x1 M<=2048 stages=4"
NR_IGEMM_FORCE="64,64,-1,4,-1,4" run "1x1 M<=2048 64x64 stages=4"
NR_IGEMM_FORCE="64,64,2,2,-1,4" run "1x1 M<=2048 64x64 split2"
export NR_IGEMM_FORCE_MAXM=512
NR_IGEMM_FORCE="64,64,2,2,-1,4" run "1x1 M<=512 64x64 split2"
NR_IGEMM_FORCE="64,64,-1,4,-1,4" run "1x1 M<=512 64x64 stages=4"
unset NR_IGEMM_FORCE_KS NR_IGEMM_FORCE_MAXM
run base2
