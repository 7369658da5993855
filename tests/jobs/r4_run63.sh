set -x
mkdir -p gpurun_out
for b in 80 96 112 128; do AB_BATCH=$b timeout -k 10 500 bash tests/micro/ab_b64.sh "opt:t64_tiles=127" > gpurun_out/r4_ab_b${b}_t64thr.txt 2>&1; echo "== $b"; cat gpurun_out/r4_ab_b${b}_t64thr.txt; done
