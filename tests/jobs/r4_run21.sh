set -x
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_server_load.py -m gpu -x -q > gpurun_out/r4_server_tests4.txt 2>&1
echo rc $?
tail -30 gpurun_out/r4_server_tests4.txt | cut -c1-300
