#!/bin/bash
# builds a variant of liblcd_hotpath.so into exp/lib_<name>.so: tools/ab_build.sh <name> [-DFLAG ...]   (only poa_kernel.hip is recompiled with the flags)
set -e
cd "$(dirname "$0")/../longcalld_amd/csrc"
name=$1; shift
make -j8 >/dev/null
mkdir -p ../../exp build_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -w "$@" -c poa_kernel.hip -o build_$name/poa_kernel.hip.o
objs=$(ls build/*.o | grep -v poa_kernel)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../exp/lib_$name.so build_$name/poa_kernel.hip.o $objs -lz -ldl
rm -rf build_$name
echo built exp/lib_$name.so
