# Whole jobs with 1 .. 16 units per GPU (bench.py --shard outputs --units-per-gpu U: fit + Sobol train-points/s, host share) at the sizes the
# reference's own sweep runs (benchmark_script.py:35-40) and at the C3 share:   gpurun -- bash tools/bench_units.sh   then tools/collect_units_bench.py
B="python bench.py --no-host-api --no-cpu-baseline --steps 3 --warmup 1 --shard outputs"
run() { $B --rows $1 --dims $2 --units-per-gpu $3 > gpurun_out/r04_bench_n$1_units$3.json 2> gpurun_out/r04_bench_n$1_units$3.err || echo FAIL $1 $2 $3; }
for U in 1 8 16; do run 512 3 $U; done
for U in 1 8 16; do run 1024 5 $U; done
for U in 1 8 16; do run 2048 10 $U; done
for U in 1 4 8 16; do run 4096 10 $U; done
for U in 1 2 4; do run 8192 10 $U; done
python bench.py --rows 8192 --dims 10 --shard outputs > gpurun_out/r04_bench_c3_share.json 2>/dev/null || echo FAIL share
python tools/config_report.py --save > gpurun_out/config_report.log 2>&1 || echo FAIL config
