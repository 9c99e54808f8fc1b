#!/bin/bash
# Kernel trace of tools/potrf_once.py N M under the current environment, analysed by a script:  gpurun -- bash tools/trace_run.sh 16384 10 tools/dloop_steps.py
N=$1; M=$2; SCRIPT=$3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_raw
rm -rf $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/potrf_once.py $N $M > $OUT.log 2>&1
F=$(find $OUT -name '*kernel_trace.csv' | head -1)
python3 $SCRIPT "$F"
rm -rf $OUT
