#!/bin/bash
# Build _ab/lib<NAME>.so from the working tree with extra hipcc flags:  tools/make_ab_variant.sh NAME -DFOO=1 ...
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$root/_ab"
src=$(python3 -c "import sys; sys.path.insert(0, '$root'); import lightloam_amd; from lightloam_amd import build as b; print(' '.join(b.HIP_SOURCES)); print(' '.join(b.HIPCC_FLAGS))")
files=$(echo "$src" | head -1); flags=$(echo "$src" | tail -1)
(cd "$root/light-loam_amd/csrc" && /opt/rocm/bin/hipcc $flags "$@" -I "$root/include" -I . -o "$root/_ab/lib$name.so" $files)
echo "built _ab/lib$name.so with $*"
