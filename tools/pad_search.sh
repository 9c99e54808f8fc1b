#!/bin/bash
# Random search over RCGP_STREAM_PAD (idle streams created between the library's six): which stream-to-hardware-queue placements are fast?
#   gpurun -- bash tools/pad_search.sh 16384 10 40
N=$1; M=$2; K=$3
python3 - "$K" > /tmp/pads.txt <<'PY'
import random, sys
random.seed(7)
seen = set()
while len(seen) < int(sys.argv[1]):
    seen.add(':'.join(str(random.choice([0, 0, 1, 2, 3])) for _ in range(6)))
print('\n'.join(sorted(seen)))
PY
while read P; do
  RCGP_STREAM_PAD=$P SWEEP_KNOBS='[{}]' timeout -k 10 120 python tools/potrf_sweep.py $N $M 2>&1 | tail -1 | sed "s/^/[$P] /"
done < /tmp/pads.txt
