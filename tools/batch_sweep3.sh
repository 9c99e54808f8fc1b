#!/bin/bash
# lean-step threshold for batches, one process per knob set. Usage: tools/batch_sweep3.sh N M units out.txt
N=$1; M=$2; U=$3; OUT=$4
: > $OUT
for L in 40 28 20 12 0; do
  RCGP_LEAN=$L python tools/batch_potrf.py $N $M $U 4 >> $OUT 2>&1 || exit 1
done
