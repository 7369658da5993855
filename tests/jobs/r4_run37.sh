set -x
mkdir -p gpurun_out
NASR_REPORT_DIR=gpurun_out/r4_reports timeout -k 10 1150 python3 -m pytest tests -m gpu -x -q > gpurun_out/r4_gpu_suite4.txt 2>&1
echo suite rc $?
tail -5 gpurun_out/r4_gpu_suite4.txt
