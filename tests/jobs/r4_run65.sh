set -x
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4_bench_final5.json 2> gpurun_out/r4_bench_final5.err
echo bench rc $?
