set -x
mkdir -p gpurun_out
for b in 128 256 512; do AB_R=0 AB_BATCH=$b timeout -k 10 300 bash tests/micro/ab_b64.sh > gpurun_out/r4_ab_b${b}_R0.txt 2>&1; echo "== $b R=0"; cat gpurun_out/r4_ab_b${b}_R0.txt; done
