#!/bin/bash
# One potrf_sweep configuration per PROCESS (stream sets -- CU-masked queues among them -- live as long as the process, so knob values that
# shape the streams must not share one):   gpurun -- bash tools/sweep_fresh.sh 16384 10 'RCGP_A=1 RCGP_B=2' 'RCGP_A=0' ...
N=$1; M=$2; shift 2
for CFG in "" "$@"; do
  env $CFG SWEEP_KNOBS='[{}, {}]' timeout -k 10 120 python tools/potrf_sweep.py $N $M 2>&1 | tail -1 | sed "s/^/[$CFG] /"
done
