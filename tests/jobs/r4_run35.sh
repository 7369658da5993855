set -x
mkdir -p gpurun_out
timeout -k 10 900 python3 tests/micro/gemm_variant_identity.py "opt:wide_tiles=0" "opt:gemm_cores=0" > gpurun_out/r4_variant_identity_order.txt 2>&1
echo identity rc $?
cat gpurun_out/r4_variant_identity_order.txt
for b in 128 256 512; do AB_BATCH=$b timeout -k 10 500 bash tests/micro/ab_b64.sh > gpurun_out/r4_ab_b${b}_order2.txt 2>&1; cat gpurun_out/r4_ab_b${b}_order2.txt; done
timeout -k 10 300 bash tests/micro/ab_b64.sh > gpurun_out/r4_ab_b64_order2.txt 2>&1; cat gpurun_out/r4_ab_b64_order2.txt
