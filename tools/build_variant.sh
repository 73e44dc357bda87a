#!/bin/bash
# Builds a library variant into build_variants/lib_<name>.so from the current sources with a replacement
# match_mfma_kernels source and extra flags: tools/build_variant.sh <name> <mfma source> [flags...]
set -e
NAME=$1; SRC=$(readlink -f $2); shift 2
ROOT=$(cd $(dirname $0)/.. && pwd)
mkdir -p $ROOT/build_variants
cd $ROOT/slam_amd/csrc
cp $SRC /tmp/variant_$NAME.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I. "$@" -c /tmp/variant_$NAME.hip -o /tmp/variant_$NAME.o
OBJS=$(ls *.o | grep -v match_mfma_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build_variants/lib_$NAME.so $OBJS /tmp/variant_$NAME.o
echo built build_variants/lib_$NAME.so
