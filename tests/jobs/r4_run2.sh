set -x
mkdir -p gpurun_out
timeout -k 10 300 python3 tests/micro/vad_bench.py > gpurun_out/r4_vad_bench2.txt 2>&1 || exit 1
timeout -k 10 900 python3 -m pytest tests/test_gpu_diar.py -m gpu -q > gpurun_out/r4_diar_tests2.txt 2>&1
echo diar tests rc $?
cd tests/micro
PROBE32=1 PROBE_PERSIST=1 PROBE_M=7168 timeout -k 10 300 ./gemm_probe > ../../gpurun_out/r4_persist_order_M7168.txt 2>&1 || exit 1
PROBE32=1 PROBE_PERSIST=1 PROBE_M=1792 timeout -k 10 300 ./gemm_probe > ../../gpurun_out/r4_persist_order_M1792.txt 2>&1 || exit 1
