#!/bin/bash
# tail length x panel width for batches, one process per knob set. Usage: tools/batch_sweep4.sh N M units out.txt
N=$1; M=$2; U=$3; OUT=$4
: > $OUT
for TN in "64 1024" "32 512" "16 512" "16 256" "8 256" "0 512"; do
  set -- $TN
  RCGP_TAIL=$1 RCGP_NB=$2 python tools/batch_potrf.py $N $M $U 4 >> $OUT 2>&1 || exit 1
done
