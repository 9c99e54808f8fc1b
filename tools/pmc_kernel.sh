#!/bin/bash
# One PMC counter for one kernel name over tools/one_eval.py, for each value of a knob (run through gpurun from the repo root):
#   gpurun -- bash tools/pmc_kernel.sh FETCH_SIZE k_grad RCGP_GRAD_ORDER 0 1
# Prints the summed counter per launch of that kernel (FETCH_SIZE / WRITE_SIZE are KB; double FETCH_SIZE for wide loads on gfx950).
set -o pipefail
CTR=$1; KERNEL=$2; KNOB=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for V in "$@"; do
  OUT=gpurun_out/pmc_raw_${CTR}_$V
  rm -rf $OUT
  export $KNOB=$V
  timeout -k 10 300 rocprofv3 --pmc $CTR --kernel-trace --output-format csv -d $OUT -- python3 tools/one_eval.py ${SIZE:-16384 10} > $OUT.log 2>&1
  F=$(find $OUT -name '*counter_collection.csv' | head -1)
  python3 - "$F" "$KERNEL" "$CTR" "$KNOB=$V" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
tot = collections.defaultdict(float); ids = collections.defaultdict(set)
for r in rows:
    k = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
    if r['Counter_Name'] == sys.argv[3]:
        tot[k] += float(r['Counter_Value']); ids[k].add(r['Dispatch_Id'])
for k in sorted(tot, key=lambda k: -tot[k])[:8]:
    mark = '  <==' if k == sys.argv[2] else ''
    print(f'{sys.argv[4]:22s} {k:22s} launches {len(ids[k]):4d}  {sys.argv[3]} per launch {tot[k] / len(ids[k]):14.1f}{mark}')
PY
  rm -rf $OUT
done
