set -x
mkdir -p gpurun_out
AB_BATCH=128 timeout -k 10 700 bash tests/micro/ab_b64.sh "opt:wide_tiles=1128" "opt:wide_tiles=1096" "opt:wide_tiles=1064" > gpurun_out/r4_ab_b128_widethr.txt 2>&1; grep pipelined gpurun_out/r4_ab_b128_widethr.txt
AB_BATCH=192 timeout -k 10 700 bash tests/micro/ab_b64.sh "opt:wide_tiles=1192" "opt:wide_tiles=1144" "opt:wide_tiles=1096" > gpurun_out/r4_ab_b192_widethr.txt 2>&1; grep pipelined gpurun_out/r4_ab_b192_widethr.txt
AB_BATCH=96 timeout -k 10 700 bash tests/micro/ab_b64.sh "opt:wide_tiles=1096" "opt:wide_tiles=1064" > gpurun_out/r4_ab_b96_widethr.txt 2>&1; grep pipelined gpurun_out/r4_ab_b96_widethr.txt
