set -x
mkdir -p gpurun_out
timeout -k 10 900 python3 tests/micro/gemm_variant_identity.py "opt:wide_tiles=256" "opt:wide_tiles=0" > gpurun_out/r4_variant_identity224.txt 2>&1
echo identity rc $?
cat gpurun_out/r4_variant_identity224.txt
cd tests/micro && timeout -k 10 400 ./persist_probe cold 3584 7168 > ../../gpurun_out/r4_wide224_probe_cold.txt 2>&1; cd ../..
cut -c1-120,196-330 gpurun_out/r4_wide224_probe_cold.txt
AB_BATCH=512 timeout -k 10 900 bash tests/micro/ab_b64.sh "opt:wide_tiles=256" > gpurun_out/r4_ab_b512_wide224.txt 2>&1
cat gpurun_out/r4_ab_b512_wide224.txt
AB_BATCH=256 timeout -k 10 600 bash tests/micro/ab_b64.sh "opt:wide_tiles=256" > gpurun_out/r4_ab_b256_wide224.txt 2>&1
cat gpurun_out/r4_ab_b256_wide224.txt
