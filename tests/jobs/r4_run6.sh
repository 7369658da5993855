set -x
mkdir -p gpurun_out
cd tests/micro && timeout -k 10 300 ./persist_probe > ../../gpurun_out/r4_persist_probe.txt 2>&1; cd ../..
cat gpurun_out/r4_persist_probe.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_server_load.py -m gpu -q -x > gpurun_out/r4_server_test.txt 2>&1
echo server test rc $?
tail -5 gpurun_out/r4_server_test.txt
for cfg in "burst 13 60" "burst 0 30" "realtime 0 20" "realtime 13 24"; do set -- $cfg
  timeout -k 10 600 python3 tests/server_load.py --streams 64 --seconds $3 --right-context $2 --mode $1 --warmup-seconds 6 > gpurun_out/r4_server_load_$1_R$2.json 2> gpurun_out/r4_server_load_$1_R$2.err
  echo load $cfg rc $?
  cut -c1-900 gpurun_out/r4_server_load_$1_R$2.json
done
timeout -k 10 900 python3 bench.py --gpus 2 --share-device 0 --steps 20 --warmup 5 --no-b512 --no-f32-engine --no-buffered --no-cpu-baseline > gpurun_out/r4_two_ranks.txt 2> gpurun_out/r4_two_ranks.err
echo two ranks rc $?
tail -c 1500 gpurun_out/r4_two_ranks.txt
