# same-box A/B of build variants: rebuild the library on the GPU box per variant, run the headline bench
cd $GRAFT_REPO_ROOT
run() {  # $1 = label, $2 = EXTRA, $3 = NOPK override ("default" keeps the Makefile's)
  make -C neurons_amd/csrc clean > /dev/null 2>&1
  if [ "$3" = "default" ]; then make -C neurons_amd/csrc -j16 EXTRA="$2" > /dev/null 2>&1; else make -C neurons_amd/csrc -j16 EXTRA="$2" NOPK="$3" > /dev/null 2>&1; fi
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-psnr 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], 'fps', d['config']['ms_per_ddim_step'], 'ms/step igemm', d['roofline']['ms_per_ddim_step'], d['roofline']['per_class_ms_per_ddim_step'])"
}
run "nopk+asm      " "" default
run "nopk+builtin  " "-DNR_GLDS_BUILTIN" default
run "pk+asm        " "" " "
run "nopk+asm again" "" default
