#!/bin/bash
# rocprofv3 evidence for a BATCHED evaluation (tools/batch_potrf_once.py N M units): kernel stats of three whole evaluations, and the
# MFMA-busy PMC pass of three batched factorisations (one counter group per run, --kernel-trace only beside it).
#   gpurun -- bash tools/profile_batch.sh 8192 10 4     ->  gpurun_out/prof_batch_<N>x<units>/ ; summaries: tools/collect_batch_profiles.py
set -o pipefail
N=$1; M=$2; U=$3
OUT=gpurun_out/prof_batch_${N}x${U}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eval -- python3 tools/batch_potrf_once.py $N $M $U 3 > $OUT/eval.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_busy -- python3 tools/batch_potrf_once.py $N $M $U 1 > $OUT/pmc_busy.log 2>&1 || exit 1
find $OUT -name '*kernel_trace.csv' -path '*eval*' -size +8M -delete
echo done >&2
