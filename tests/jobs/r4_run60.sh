set -x
mkdir -p gpurun_out
AB_SYNC=0 timeout -k 10 600 bash tests/micro/ab_b64.sh "opt:tile_bands=16" "opt:tile_bands=4" "opt:tile_bands=2" > gpurun_out/r4_ab_b64_bandw.txt 2>&1; cat gpurun_out/r4_ab_b64_bandw.txt
AB_SYNC=0 AB_BATCH=512 timeout -k 10 600 bash tests/micro/ab_b64.sh "opt:tile_bands=16" "opt:tile_bands=4" > gpurun_out/r4_ab_b512_bandw.txt 2>&1; cat gpurun_out/r4_ab_b512_bandw.txt
