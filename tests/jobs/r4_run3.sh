set -x
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests/test_gpu_diar.py -m gpu -q > gpurun_out/r4_diar_tests3.txt 2>&1
echo diar tests rc $?
cd tests/micro
PROBE_STAMPS=1 PROBE32=1 PROBE_PERSIST=1 PROBE_M=7168 timeout -k 10 300 ./gemm_probe > ../../gpurun_out/r4_persist_stamps_M7168.txt 2>&1 || exit 1
