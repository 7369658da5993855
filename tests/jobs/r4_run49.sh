set -x
mkdir -p gpurun_out
AB_SYNC=0 timeout -k 10 600 bash tests/micro/ab_b64.sh "opt:wide_tiles=2128" "opt:wide_tiles=2096" "opt:wide_tiles=2064" > gpurun_out/r4_ab_b64_w128.txt 2>&1; cat gpurun_out/r4_ab_b64_w128.txt
