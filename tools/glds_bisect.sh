# race bisect: rebuild with a variant flag on the GPU box, then repeat the mismatch hunt
cd $GRAFT_REPO_ROOT
for v in "-fno-slp-vectorize" ""; do
  rm -f neurons_amd/csrc/build/norm.o
  make -C neurons_amd/csrc EXTRA="$v" > /dev/null 2>&1
  echo "variant [$v]"
  python tools/race_hunt.py default default 2>&1 | grep -v amdgpu.ids
done
