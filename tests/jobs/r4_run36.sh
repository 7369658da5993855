set -x
mkdir -p gpurun_out
timeout -k 10 500 bash tests/micro/ab_b64.sh "opt:tile_bands=1" > gpurun_out/r4_ab_b64_bands.txt 2>&1; cat gpurun_out/r4_ab_b64_bands.txt
AB_BATCH=128 timeout -k 10 500 bash tests/micro/ab_b64.sh "opt:tile_bands=0" > gpurun_out/r4_ab_b128_bands.txt 2>&1; cat gpurun_out/r4_ab_b128_bands.txt
AB_BATCH=32 timeout -k 10 500 bash tests/micro/ab_b64.sh "opt:tile_bands=1" > gpurun_out/r4_ab_b32_bands.txt 2>&1; cat gpurun_out/r4_ab_b32_bands.txt
