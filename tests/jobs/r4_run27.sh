set -x
mkdir -p gpurun_out
timeout -k 10 1000 bash tests/micro/ab_b64.sh "opt:deep_k_splits=2" "opt:deep_k_splits=4" "opt:deep_k_splits=1" > gpurun_out/r4_ab_deepk.txt 2>&1
cat gpurun_out/r4_ab_deepk.txt
