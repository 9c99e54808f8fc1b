#!/bin/bash
# One potrf_sweep process per environment string:   gpurun -- bash tools/env_search.sh 16384 10 'A=1 B=2' 'C=3' ...
N=$1; M=$2; shift 2
for CFG in "$@"; do
  env $CFG SWEEP_KNOBS='[{}]' timeout -k 10 120 python tools/potrf_sweep.py $N $M 2>&1 | tail -1 | sed "s/^/[$CFG] /"
done
