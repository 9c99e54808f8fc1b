#!/bin/bash
# far-update grouping and the half-tile threshold of the L^-1 levels, one process per knob set. Usage: tools/batch_sweep2.sh N M units out.txt
N=$1; M=$2; U=$3; OUT=$4
: > $OUT
for G in 2 3 4 6; do
  for T in 64 32; do
    RCGP_FARG=$G RCGP_TAIL=$T RCGP_NB=512 python tools/batch_potrf.py $N $M $U 4 >> $OUT 2>&1 || exit 1
  done
done
for H in 0 512 2048; do
  RCGP_HALF_TILES=$H python tools/batch_potrf.py $N $M $U 4 >> $OUT 2>&1 || exit 1
done
