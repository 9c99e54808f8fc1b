#!/bin/bash
# The C2 factorisation under different creation orders of the library's streams (RCGP_STREAM_ORDER), one process each:
#   gpurun -- bash tools/order_search.sh 16384 10 025634 02563 302564 ...
N=$1; M=$2; shift 2
for O in "$@"; do
  RCGP_STREAM_ORDER=$O SWEEP_KNOBS='[{}]' timeout -k 10 120 python tools/potrf_sweep.py $N $M 2>&1 | tail -1 | sed "s/^/[$O] /"
done
