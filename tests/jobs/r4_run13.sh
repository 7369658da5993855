set -x
mkdir -p gpurun_out
AB_BATCH=512 timeout -k 10 900 bash tests/micro/ab_b64.sh "opt:wide_tiles=0" > gpurun_out/r4_ab_b512_wide2.txt 2>&1
AB_BATCH=256 timeout -k 10 600 bash tests/micro/ab_b64.sh "opt:wide_tiles=0" > gpurun_out/r4_ab_b256_wide2.txt 2>&1
AB_BATCH=128 timeout -k 10 600 bash tests/micro/ab_b64.sh "opt:wide_tiles=0" > gpurun_out/r4_ab_b128_wide2.txt 2>&1
cat gpurun_out/r4_ab_b512_wide2.txt gpurun_out/r4_ab_b256_wide2.txt gpurun_out/r4_ab_b128_wide2.txt
timeout -k 10 300 python3 tests/micro/diar_bench.py > gpurun_out/r4_diar_bench3.txt 2>&1
timeout -k 10 300 python3 tests/micro/diar_bench.py >> gpurun_out/r4_diar_bench3.txt 2>&1
cat gpurun_out/r4_diar_bench3.txt
timeout -k 10 600 python3 -m pytest tests/test_gpu_speech.py -m gpu -q -x -k harder > gpurun_out/r4_harder_test.txt 2>&1
echo harder rc $?
tail -5 gpurun_out/r4_harder_test.txt
