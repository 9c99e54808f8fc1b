#!/bin/bash
# Stand-alone Cholesky time of the shipped library and of dev builds (tools/build_dev_lib.sh), one process per (library, size set):
#   gpurun -- bash tools/dev_sweep.sh "4096 5 8192 5 16384 10" default tools/dev/a.so tools/dev/b.so
SIZES=$1; shift
for LIB in "$@"; do
  if [ "$LIB" = default ]; then unset RCGP_DEV_LIB; else export RCGP_DEV_LIB=$LIB; fi
  echo "== $LIB"
  SWEEP_KNOBS='[{}]' timeout -k 10 300 python3 tools/potrf_sweep.py $SIZES 2>&1 | grep -v "^$"
done
