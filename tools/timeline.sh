#!/bin/bash
# Kernel-trace timeline of the Cholesky's panel chain at one size, for each value of a knob (run through gpurun from the repo root):
#   gpurun -- bash tools/timeline.sh 4096 5 RCGP_NB 512 1024
# Writes gpurun_out/tl_<N>_<knob><value>.txt (tools/chain_timeline.py on the trace of tools/potrf_once.py).
set -o pipefail
N=$1; M=$2; KNOB=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for V in "$@"; do
  OUT=gpurun_out/tl_raw_$V
  rm -rf $OUT
  env_line="$KNOB=$V"
  export $KNOB=$V
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/potrf_once.py $N $M > $OUT.log 2>&1
  F=$(find $OUT -name '*kernel_trace.csv' | head -1)
  python3 tools/chain_timeline.py "$F" ${ROWS:-70} > gpurun_out/tl_${N}_${KNOB}${V}.txt 2>&1
  rm -rf $OUT
done
head -14 gpurun_out/tl_${N}_${KNOB}$1.txt
