set -x
mkdir -p gpurun_out
for b in 80 96 112; do AB_BATCH=$b timeout -k 10 500 bash tests/micro/ab_b64.sh "opt:gemm_cores=1" > gpurun_out/r4_ab_b${b}_cores.txt 2>&1; echo "== $b"; grep sync gpurun_out/r4_ab_b${b}_cores.txt; done
