#!/bin/bash
# One process per knob set (the knobs are read once per process): stage times of a batch of units. Usage: tools/batch_sweep.sh N M units out.txt
N=$1; M=$2; U=$3; OUT=$4
: > $OUT
for T in 64 32 16 0; do
  for NB in 512 1024; do
    for EXT in 2 4; do
      RCGP_TAIL=$T RCGP_NB=$NB RCGP_EXT=$EXT python tools/batch_potrf.py $N $M $U 4 >> $OUT 2>&1 || exit 1
    done
  done
done
