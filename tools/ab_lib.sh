#!/bin/bash
# Builds the library of another commit as neurons_amd/libneurons_amd_base.so (NR_LIB_VARIANT=base), so that one gpurun call can time two
# builds on the SAME device (MI355X boxes differ by several per cent; an A/B across two calls measures the box).
# Usage: tools/ab_lib.sh [git-rev, default HEAD]
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/nr_base.XXXX)
git -C "$root" archive "$rev" neurons_amd/csrc include | tar -x -C "$tmp"
make -C "$tmp/neurons_amd/csrc" -j8 > "$tmp/build.log" 2>&1 || { tail -n 20 "$tmp/build.log"; exit 1; }
cp "$tmp/neurons_amd/libneurons_amd.so" "$root/neurons_amd/libneurons_amd_base.so"
rm -rf "$tmp"
echo "built $root/neurons_amd/libneurons_amd_base.so from $rev"
