#!/bin/bash
# Build _ab/libA.so (the "A" side of tools/ab_bench.sh) from a git revision (default HEAD), with the flags of light-loam_amd/build.py.
set -e
rev=${1:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$rev" light-loam_amd/csrc include | tar -x -C "$tmp"
mkdir -p "$root/_ab"
src=$(python3 -c "import sys; sys.path.insert(0, '$root'); import lightloam_amd; from lightloam_amd import build as b; print(' '.join(b.HIP_SOURCES)); print(' '.join(b.HIPCC_FLAGS))")
files=$(echo "$src" | head -1); flags=$(echo "$src" | tail -1)
(cd "$tmp/light-loam_amd/csrc" && /opt/rocm/bin/hipcc $flags -I "$tmp/include" -I . -o "$root/_ab/libA.so" $files)
rm -rf "$tmp"
echo "built _ab/libA.so from $rev"
