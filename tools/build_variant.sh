#!/bin/bash
# usage: tools/build_variant.sh <name: chess|decimate|cc|preprocess> <variant.hip> <out.so> [extra flags]
set -e
R=$(cd $(dirname $0)/.. && pwd)
NAME=$1; SRC=$2; OUT=$3; shift 3
B=$R/mrgingham_amd/csrc/build
make -s -C $R/mrgingham_amd/csrc -j4 all >/dev/null
cp $SRC $R/mrgingham_amd/csrc/_variant_$NAME.hip
EXTRA=""; [ "$NAME" = chess ] && EXTRA="-mllvm -amdgpu-sched-strategy=max-ilp"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -Wno-unused-value -DMRG_EXPERIMENT $EXTRA "$@" -c $R/mrgingham_amd/csrc/_variant_$NAME.hip -o /tmp/_variant_$NAME.o
rm -f $R/mrgingham_amd/csrc/_variant_$NAME.hip
OBJS=""
for n in chess chess16 decimate preprocess preprocess16 cc blobs api; do if [ $n = $NAME ]; then OBJS="$OBJS /tmp/_variant_$NAME.o"; else OBJS="$OBJS $B/$n.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS $B/grid.o $B/image_io.o -lz
echo built $OUT
