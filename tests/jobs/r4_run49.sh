set -x
mkdir -p gpurun_out
AB_SYNC=0 timeout -k 10 400 bash tests/micro/ab_b64.sh "opt:wide_tiles=3" > gpurun_out/r4_ab_b64_rule64.txt 2>&1; cat gpurun_out/r4_ab_b64_rule64.txt
