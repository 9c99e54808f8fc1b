#!/bin/bash
# Kernel-trace timeline of the Cholesky's panel chain at one size for the shipped library and for dev builds (tools/build_dev_lib.sh):
#   gpurun -- bash tools/timeline_lib.sh 4096 5 default tools/dev/ntk128.so
# Writes gpurun_out/tl_<N>_<name>.txt (tools/chain_timeline.py on the trace of tools/potrf_once.py).
set -o pipefail
N=$1; M=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for LIB in "$@"; do
  NAME=$(basename $LIB .so)
  OUT=gpurun_out/tl_raw_$NAME
  rm -rf $OUT
  if [ "$LIB" = default ]; then unset RCGP_DEV_LIB; else export RCGP_DEV_LIB=$LIB; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/potrf_once.py $N $M > $OUT.log 2>&1
  F=$(find $OUT -name '*kernel_trace.csv' | head -1)
  python3 tools/chain_timeline.py "$F" ${ROWS:-60} > gpurun_out/tl_${N}_${NAME}.txt 2>&1
  rm -rf $OUT
done
