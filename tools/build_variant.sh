#!/bin/bash
# Builds a library variant into build_variants/lib_<name>.so: the current objects of slam_amd/csrc with one source
# recompiled with extra flags (or replaced by another file of the same role):
#   tools/build_variant.sh <name> <source in slam_amd/csrc, or a replacement for the file of that name> [flags...]
set -e
NAME=$1; SRC=$(readlink -f $2); shift 2
ROOT=$(cd $(dirname $0)/.. && pwd)
BASE=$(basename $SRC .hip)
mkdir -p $ROOT/build_variants
cd $ROOT/slam_amd/csrc
EXTRA=""
{ [ $BASE = match_mfma_kernels ] || [ $BASE = extract_kernels ]; } && EXTRA="-mllvm -amdgpu-mfma-vgpr-form"
cp $SRC ./variant_tmp_$NAME.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off $EXTRA -I. "$@" -c ./variant_tmp_$NAME.hip -o /tmp/variant_$NAME.o
rm -f ./variant_tmp_$NAME.hip
OBJS=$(ls *.o | grep -v "^$BASE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build_variants/lib_$NAME.so $OBJS /tmp/variant_$NAME.o
echo built build_variants/lib_$NAME.so
