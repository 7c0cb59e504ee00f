#!/bin/bash
# Build libpmx_hip.so of the WORKING TREE with extra device-compile flags into pharmsol_amd/lib/ab/<name>.so:
#   tools/ab_variant.sh <name> "-DPMX_DYN_WAVES=3 ...";  PMX_LIB=$PWD/pharmsol_amd/lib/ab/<name>.so python bench.py ...
set -e
name=$1; flags=$2
tmp=$(mktemp -d)
cp -r Makefile include tools pharmsol_amd oracle "$tmp"/
rm -rf "$tmp/pharmsol_amd/csrc/build" "$tmp/pharmsol_amd/lib"
make -C "$tmp" pharmsol_amd/lib/libpmx_hip.so DEVFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-parameter -Iinclude $flags" >/dev/null
mkdir -p pharmsol_amd/lib/ab
cp "$tmp/pharmsol_amd/lib/libpmx_hip.so" "pharmsol_amd/lib/ab/$name.so"
rm -rf "$tmp"
echo "built pharmsol_amd/lib/ab/$name.so with $flags"
