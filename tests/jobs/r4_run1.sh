set -x
mkdir -p gpurun_out
cd tests/micro
for pm in 0 1 3 5; do timeout -k 10 120 ./cores_probe 0 1024 4096 0 0 $pm >> ../../gpurun_out/r4_prio_probe.txt 2>&1 || exit 1; done
cd ../..
timeout -k 10 300 python3 tests/micro/vad_bench.py > gpurun_out/r4_vad_bench.txt 2>&1 || exit 1
timeout -k 10 900 python3 -m pytest tests/test_gpu_diar.py -m gpu -x -q > gpurun_out/r4_diar_tests.txt 2>&1
echo diar tests rc $?
