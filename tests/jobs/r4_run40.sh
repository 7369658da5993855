set -x
mkdir -p gpurun_out
AB_CHECKPOINT=speech AB_BATCH=512 timeout -k 10 900 bash tests/micro/ab_b64.sh "opt:tile_bands=0" > gpurun_out/r4_ab_b512_speech.txt 2>&1; cat gpurun_out/r4_ab_b512_speech.txt
