set -x
mkdir -p gpurun_out
AB_BATCH=512 timeout -k 10 1100 bash tests/micro/ab_b64.sh "opt:persistent_gemm=1" "opt:persistent_gemm=1 opt:wide_tiles=0" > gpurun_out/r4_ab_b512_persist2.txt 2>&1
cat gpurun_out/r4_ab_b512_persist2.txt
