#!/bin/bash
# A/B aid: another build of the library with extra flags for ONE source file -> gpc_amd/lib/libgpc_hip_<name>.so
# (loaded by GPC_LIB_VARIANT=<name>; tools only).  usage: tools/build_variant.sh <name> <file.hip> <flags...>
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../gpc_amd/csrc"
make -s >/dev/null
mkdir -p ../../build/variants
obj=../../build/variants/${src%.hip}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Wall -Wno-unused-function "$@" -c $src -o $obj
objs=$(ls ../../build/csrc/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libgpc_hip_$name.so $objs $obj -ldl -lpthread
echo built ../lib/libgpc_hip_$name.so
