#!/bin/bash
# Builds a variant of libspiht_hip.so with extra compiler flags into build/var_<name>/spiht_amd/libspiht_hip.so
# (for A/B runs: SPIHT_HIP_LIB=build/var_<name>/spiht_amd/libspiht_hip.so python bench.py ...).
#   tools/build_variant.sh <name> <extra flags...>
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
d=$root/build/var_$name
rm -rf "$d"; mkdir -p "$d/spiht_amd" "$d/tools"
cp -r "$root/spiht_amd/csrc" "$d/spiht_amd/csrc"
cp -r "$root/include" "$d/include"
cp "$root/tools/hop_align.py" "$d/tools/"
rm -f "$d"/spiht_amd/csrc/*.o
make -C "$d/spiht_amd/csrc" -j4 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $*" >/dev/null
rm -f "$d"/spiht_amd/csrc/*.o
echo "$d/spiht_amd/libspiht_hip.so"
