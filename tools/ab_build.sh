#!/bin/bash
# Build libpmx_hip.so of another commit into pharmsol_amd/lib/ab/<name>.so for same-device A/B runs:
#   tools/ab_build.sh <commit> <name>;  PMX_LIB=pharmsol_amd/lib/ab/<name>.so python bench.py ...
set -e
commit=$1; name=$2
tmp=$(mktemp -d)
git archive "$commit" | tar -x -C "$tmp"
make -C "$tmp" pharmsol_amd/lib/libpmx_hip.so >/dev/null
mkdir -p pharmsol_amd/lib/ab
cp "$tmp/pharmsol_amd/lib/libpmx_hip.so" "pharmsol_amd/lib/ab/$name.so"
rm -rf "$tmp"
echo "built pharmsol_amd/lib/ab/$name.so from $commit"
