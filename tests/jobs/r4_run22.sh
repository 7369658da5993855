set -x
mkdir -p gpurun_out
timeout -k 10 400 python3 tests/micro/server_mixed_debug.py > gpurun_out/r4_mixed_debug.txt 2>&1
echo rc $?
tail -120 gpurun_out/r4_mixed_debug.txt | cut -c1-250
