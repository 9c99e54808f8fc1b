#!/bin/bash
# Profiles of the C2 workload on the GPU box (run through gpurun from the repo root); raw output under gpurun_out/prof_<tag>/,
# summaries are copied into profiles/ by tools/collect_profiles.py.
#   gpurun -- bash tools/profile_c2.sh r01
set -e -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# 1. the bench command itself under rocprofv3 --kernel-trace --stats
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 bench.py > $OUT/bench_under_rocprof.json 2> $OUT/bench.err
find $OUT/bench -name '*kernel_trace.csv' -delete      # (tens of MB: gpurun brings back at most 64 MiB; the stats csv beside it is what is kept)
echo "bench under rocprof done" >&2
# 2. one LML+gradient evaluation + Sobol at C2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/eval -- python3 tools/one_eval.py > $OUT/eval.log 2>&1
# 3. PMC passes (one counter group per run; no other trace domains)
for C in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  timeout -k 10 300 rocprofv3 --pmc $C GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 tools/one_eval.py > $OUT/pmc_$C.log 2>&1
  echo "pmc $C done" >&2
done
# 4. the un-profiled bench line
timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench_plain.err
tail -c 400 $OUT/bench.json
