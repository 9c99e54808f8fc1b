set -o pipefail
B="python bench.py --no-host-api --no-cpu-baseline --steps 2 --warmup 1 --shard outputs"
$B --rows 8192 --dims 10 --units-per-gpu 1 > gpurun_out/r04_bench_c3_units1.json 2>/dev/null || echo F1
$B --rows 8192 --dims 10 --units-per-gpu 2 > gpurun_out/r04_bench_c3_units2.json 2>/dev/null || echo F2
$B --rows 8192 --dims 10 --units-per-gpu 4 > gpurun_out/r04_bench_c3_units4.json 2>/dev/null || echo F3
for U in 1 8 16; do $B --rows 2048 --dims 10 --units-per-gpu $U > gpurun_out/r04_bench_n2048_units$U.json 2>/dev/null || echo F2048_$U; done
for U in 1 8 16; do $B --rows 512 --dims 7 --units-per-gpu $U > gpurun_out/r04_bench_n512_units$U.json 2>/dev/null || echo F512_$U; done
for U in 1 4 8; do $B --rows 4096 --dims 10 --units-per-gpu $U > gpurun_out/r04_bench_n4096_units$U.json 2>/dev/null || echo F4096_$U; done
python bench.py --rows 8192 --dims 10 --shard outputs > gpurun_out/r04_bench_c3_share.json 2>/dev/null || echo Fshare
python tools/config_report.py --save > gpurun_out/config_report.log 2>&1 || echo Fcfg
