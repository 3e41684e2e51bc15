#!/bin/bash
# Builds libkmc_<name>.so from the current sources with extra compiler flags and/or another
# kmc_walk.hip.h (same-box A/B runs of kernel variants: KMC_LIB_PATH selects the library).
# usage: tools/build_variant.sh <name> "<extra hipcc flags>" [walk header]
set -e
name=$1; flags=$2; hdr=${3:-}
root=$(cd "$(dirname "$0")/.." && pwd)
w=/tmp/kmc_var_$name
rm -rf $w && mkdir -p $w/pkg/csrc $w/include
cp $root/k-mer-count_amd/csrc/* $w/pkg/csrc/ && cp $root/include/kmc.h $w/include/
[ -n "$hdr" ] && cp $hdr $w/pkg/csrc/kmc_walk.hip.h
cd $w/pkg
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I../include $flags -c csrc/kmc_api.hip -o kmc_api.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $root/k-mer-count_amd/libkmc_$name.so kmc_api.o $root/k-mer-count_amd/kmc_host.o
ls -la $root/k-mer-count_amd/libkmc_$name.so
