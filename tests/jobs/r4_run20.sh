set -x
mkdir -p gpurun_out
cd tests/micro && timeout -k 10 400 ./persist_probe 3584 7168 15360 > ../../gpurun_out/r4_wide_pf_hot.txt 2>&1; timeout -k 10 400 ./persist_probe cold 3584 7168 > ../../gpurun_out/r4_wide_pf_cold.txt 2>&1; cd ../..
grep -E "W1|pw1|spk|W2" gpurun_out/r4_wide_pf_hot.txt gpurun_out/r4_wide_pf_cold.txt | awk -F'|' '{print $1 "|" $4 "|" $5}'
timeout -k 10 900 python3 tests/micro/gemm_variant_identity.py "opt:wide_tiles=0" "opt:wide_tiles=2" > gpurun_out/r4_variant_identity4.txt 2>&1
cat gpurun_out/r4_variant_identity4.txt
AB_BATCH=512 timeout -k 10 900 bash tests/micro/ab_b64.sh "opt:wide_tiles=2" > gpurun_out/r4_ab_b512_pf.txt 2>&1
cat gpurun_out/r4_ab_b512_pf.txt
