set -x
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_host_gguf.py tests/test_gpu_server_load.py -m gpu -x -q > gpurun_out/r4_host_tests2.txt 2>&1
echo host tests rc $?
tail -5 gpurun_out/r4_host_tests2.txt
for cfg in "burst 13 120" "burst 0 120" "realtime 0 20" "realtime 13 24"; do set -- $cfg
  timeout -k 10 600 python3 tests/server_load.py --streams 64 --seconds $3 --right-context $2 --mode $1 --warmup-seconds 6 > gpurun_out/r4_server_load_$1_R$2_c.json 2> gpurun_out/r4_server_load_$1_R$2_c.err
  echo load $cfg rc $?
  cut -c1-330 gpurun_out/r4_server_load_$1_R$2_c.json
done
(time timeout -k 10 900 python3 bench.py --steps 20 --warmup 5) > gpurun_out/r4_bench_default_3.txt 2> gpurun_out/r4_bench_default_3.err
tail -c 1200 gpurun_out/r4_bench_default_3.txt
tail -4 gpurun_out/r4_bench_default_3.err
