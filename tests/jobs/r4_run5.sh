set -x
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "f32_mfma or fallback" > gpurun_out/r4_f32_tests2.txt 2>&1
echo f32 tests rc $?
timeout -k 10 900 python3 tests/micro/gemm_variant_identity.py "opt:persistent_gemm=0" "opt:gemm_cores=0 opt:persistent_gemm=0" > gpurun_out/r4_variant_identity.txt 2>&1
echo identity rc $?
cat gpurun_out/r4_variant_identity.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_diar.py -m gpu -q -x > gpurun_out/r4_diar_tests5.txt 2>&1
echo diar rc $?
AB_BATCH=512 timeout -k 10 900 bash tests/micro/ab_b64.sh "opt:persistent_gemm=0" > gpurun_out/r4_ab_b512_persist.txt 2>&1
AB_BATCH=128 timeout -k 10 600 bash tests/micro/ab_b64.sh "opt:persistent_gemm=0" > gpurun_out/r4_ab_b128_persist.txt 2>&1
timeout -k 10 300 python3 tests/micro/diar_bench.py > gpurun_out/r4_diar_bench.txt 2>&1
cat gpurun_out/r4_ab_b512_persist.txt gpurun_out/r4_ab_b128_persist.txt
