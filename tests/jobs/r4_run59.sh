set -x
mkdir -p gpurun_out
AB_CHECKPOINT=speech timeout -k 10 600 bash tests/micro/ab_b64.sh "opt:tile_bands=0" "opt:tile_bands=0 opt:wide_tiles=0" > gpurun_out/r4_ab_b64_speech_bands.txt 2>&1; cat gpurun_out/r4_ab_b64_speech_bands.txt
