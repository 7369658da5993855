set -x
mkdir -p gpurun_out
timeout -k 10 1150 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_gpu_suite.txt 2>&1
echo suite rc $?
tail -15 gpurun_out/r4_gpu_suite.txt
