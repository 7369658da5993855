set -x
mkdir -p gpurun_out
for cfg in "burst 13 120" "burst 0 120" "realtime 0 20" "realtime 13 24"; do set -- $cfg
  timeout -k 10 600 python3 tests/server_load.py --streams 64 --seconds $3 --right-context $2 --mode $1 --warmup-seconds 6 > gpurun_out/r4_server_load_$1_R$2_d.json 2> gpurun_out/r4_server_load_$1_R$2_d.err
  echo load $cfg rc $?
  cut -c1-330 gpurun_out/r4_server_load_$1_R$2_d.json
done
timeout -k 10 600 python3 -m pytest tests/test_gpu_server_load.py -m gpu -x -q > gpurun_out/r4_host_tests3.txt 2>&1
echo rc $?
tail -3 gpurun_out/r4_host_tests3.txt
