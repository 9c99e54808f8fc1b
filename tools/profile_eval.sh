#!/bin/bash
# rocprofv3 evidence for ONE LML+gradient evaluation + Sobol pass (tools/one_eval.py) at any size: kernel stats and the three PMC passes
# (one counter group per run, --kernel-trace only beside them). Run through gpurun from the repo root:
#   gpurun -- bash tools/profile_eval.sh r02_c4 28672 20
# then   python tools/collect_eval_profiles.py r02_c4   copies the summaries into profiles/.
set -o pipefail
TAG=$1; N=$2; M=$3
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eval -- python3 tools/one_eval.py $N $M > $OUT/eval.log 2>&1 || exit 1
echo "eval stats done" >&2
for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  timeout -k 10 400 rocprofv3 --pmc $C GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 tools/one_eval.py $N $M > $OUT/pmc_$C.log 2>&1 || exit 1
  echo "pmc $C done" >&2
done
# keep what travels back small: the per-dispatch traces are large
find $OUT -name '*kernel_trace.csv' -path '*eval*' -size +20M -delete
tail -2 $OUT/eval.log
