# The round-4 evidence on the final build, in one gpurun call (see tools/README.md):  gpurun --timeout 1150 -- bash tools/final_r04.sh
bash tools/profile_c2.sh r04 || echo F_profile_c2
bash tools/profile_batch.sh 8192 10 4 || echo F_pb1
bash tools/profile_batch.sh 4096 10 16 || echo F_pb2
for a in "256 3 16" "1024 5 16" "2048 10 16" "4096 10 16" "8192 10 4" "16384 10 2"; do set -- $a; python tools/batch_check.py $1 $2 $3 > gpurun_out/r04_batch_$1.json 2> gpurun_out/r04_batch_$1.err || echo FAIL $a; done
