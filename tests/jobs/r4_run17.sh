set -x
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_host_gguf.py tests/test_gpu_server_load.py -m gpu -x -q > gpurun_out/r4_host_tests.txt 2>&1
echo host tests rc $?
tail -5 gpurun_out/r4_host_tests.txt
for i in 1 2; do (time timeout -k 10 900 python3 bench.py --steps 20 --warmup 5) > gpurun_out/r4_bench_default_$i.txt 2> gpurun_out/r4_bench_default_$i.err; cp gpurun_out/bench_details.json gpurun_out/r4_bench_details_$i.json; tail -3 gpurun_out/r4_bench_default_$i.err; done
timeout -k 10 900 python3 bench.py --steps 200 --warmup 10 --no-b512 --no-f32-engine --no-extra-configs --no-cpu-baseline --no-buffered > gpurun_out/r4_bench_200.txt 2>/dev/null
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4_bench_default_1.txt gpurun_out/r4_bench_default_2.txt gpurun_out/r4_bench_200.txt | head
