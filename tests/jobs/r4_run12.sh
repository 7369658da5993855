set -x
mkdir -p gpurun_out
cd tests/micro && timeout -k 10 400 ./persist_probe 1792 3584 7168 15360 > ../../gpurun_out/r4_wide_probe_hot.txt 2>&1; timeout -k 10 400 ./persist_probe cold 1792 3584 7168 > ../../gpurun_out/r4_wide_probe_cold.txt 2>&1; cd ../..
cat gpurun_out/r4_wide_probe_hot.txt gpurun_out/r4_wide_probe_cold.txt
timeout -k 10 900 python3 tests/micro/gemm_variant_identity.py "opt:wide_tiles=0" "opt:persistent_gemm=1 opt:wide_tiles=0" > gpurun_out/r4_variant_identity3.txt 2>&1
cat gpurun_out/r4_variant_identity3.txt
AB_BATCH=512 timeout -k 10 900 bash tests/micro/ab_b64.sh "opt:wide_tiles=0" > gpurun_out/r4_ab_b512_wide.txt 2>&1
AB_BATCH=128 timeout -k 10 600 bash tests/micro/ab_b64.sh "opt:wide_tiles=0" > gpurun_out/r4_ab_b128_wide.txt 2>&1
AB_BATCH=256 timeout -k 10 600 bash tests/micro/ab_b64.sh "opt:wide_tiles=0" > gpurun_out/r4_ab_b256_wide.txt 2>&1
cat gpurun_out/r4_ab_b512_wide.txt gpurun_out/r4_ab_b128_wide.txt gpurun_out/r4_ab_b256_wide.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_diar.py -m gpu -q -x > gpurun_out/r4_diar_tests12.txt 2>&1
echo diar rc $?
timeout -k 10 300 python3 tests/micro/diar_bench.py > gpurun_out/r4_diar_bench2.txt 2>&1
cat gpurun_out/r4_diar_bench2.txt
