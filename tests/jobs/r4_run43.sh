set -x
mkdir -p gpurun_out
timeout -k 10 600 python3 tests/micro/gemm_variant_identity.py > gpurun_out/r4_variant_identity_nseg.txt 2>&1
echo identity rc $?; cat gpurun_out/r4_variant_identity_nseg.txt
for b in 512 256; do AB_BATCH=$b timeout -k 10 500 bash tests/micro/ab_b64.sh > gpurun_out/r4_ab_b${b}_nseg.txt 2>&1; cat gpurun_out/r4_ab_b${b}_nseg.txt; done
AB_CHECKPOINT=speech AB_BATCH=512 timeout -k 10 500 bash tests/micro/ab_b64.sh > gpurun_out/r4_ab_b512_speech_nseg.txt 2>&1; cat gpurun_out/r4_ab_b512_speech_nseg.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "pipeline or soak or identity" > gpurun_out/r4_pipe_tests.txt 2>&1; tail -3 gpurun_out/r4_pipe_tests.txt
