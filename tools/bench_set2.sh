B="python bench.py --no-host-api --no-cpu-baseline --steps 2 --warmup 1 --shard outputs"
for U in 1 8 16; do $B --rows 512 --dims 3 --units-per-gpu $U > gpurun_out/r04_bench_n512_units$U.json 2>gpurun_out/r04_bench_n512_units$U.err || echo F512_$U; done
for U in 1 16; do $B --rows 1024 --dims 5 --units-per-gpu $U > gpurun_out/r04_bench_n1024_units$U.json 2>gpurun_out/r04_bench_n1024_units$U.err || echo F1024_$U; done
python bench.py > gpurun_out/r04_bench_c2_b.json 2> gpurun_out/r04_bench_c2_b.err || echo Fc2
bash tools/profile_eval.sh r04_c1 8192 5 || echo Fc1
