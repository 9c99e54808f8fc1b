#!/bin/bash
# Kernel-trace timeline of the batched Cholesky:  gpurun -- bash tools/batch_timeline.sh N M units [stage]  ->  gpurun_out/btl_<N>_<units>.txt
set -o pipefail
N=$1; M=$2; U=$3; ST=${4:-1}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/btl_raw_${N}_${U}
rm -rf $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 tools/batch_potrf_once.py $N $M $U $ST > $OUT.log 2>&1
F=$(find $OUT -name '*kernel_trace.csv' | head -1)
python3 tools/chain_timeline.py "$F" ${ROWS:-70} > gpurun_out/btl_${N}_${U}_s${ST}.txt 2>&1
rm -rf $OUT
