#!/bin/bash
# Dev build of the library with extra compile flags into tools/dev/<name>.so (never the shipped build):
#   bash tools/build_dev_lib.sh timing -DRC_DIAG_TIMING
# Tools pick it up through RCGP_DEV_LIB=tools/dev/<name>.so (tools/diag_timing.py, tools/diag_spans.py).
set -e
NAME=$1; shift
cd "$(dirname "$0")/../rom-comma_amd/csrc"
OUT=../../tools/dev
mkdir -p $OUT/obj_$NAME
for f in api gemm gram potrf solve sobol; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -Wno-unused-value "$@" -c $f.hip -o $OUT/obj_$NAME/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OUT/obj_$NAME/*.o -o $OUT/$NAME.so
echo built $OUT/$NAME.so
