#!/bin/bash
# Panel-level timeline of one Cholesky (tools/panel_timeline.py on a kernel trace of tools/potrf_once.py), plus every kernel of a
# time window:   gpurun -- bash tools/panel_tl.sh 16384 10 [t0_us t1_us]
set -o pipefail
N=$1; M=$2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/ptl_raw
rm -rf $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/potrf_once.py $N $M > $OUT.log 2>&1
F=$(find $OUT -name '*kernel_trace.csv' | head -1)
python3 tools/panel_timeline.py "$F" > gpurun_out/panel_tl_$N.txt 2>&1
python3 tools/chain_timeline.py "$F" 10 > gpurun_out/chain_tl_$N.txt 2>&1
if [ -n "$3" ]; then python3 tools/window_dump.py "$F" $3 $4 > gpurun_out/window_$N.txt 2>&1; fi
rm -rf $OUT
cat gpurun_out/panel_tl_$N.txt
