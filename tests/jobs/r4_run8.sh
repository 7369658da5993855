set -x
mkdir -p gpurun_out
cd tests/micro && timeout -k 10 300 ./persist_probe cold 3584 7168 > ../../gpurun_out/r4_persist_probe_cold2.txt 2>&1; timeout -k 10 300 ./persist_probe 7168 > ../../gpurun_out/r4_persist_probe_hot2.txt 2>&1; cd ../..
cat gpurun_out/r4_persist_probe_cold2.txt gpurun_out/r4_persist_probe_hot2.txt
timeout -k 10 900 python3 tests/micro/gemm_variant_identity.py "opt:persistent_gemm=0" > gpurun_out/r4_variant_identity2.txt 2>&1
cat gpurun_out/r4_variant_identity2.txt
AB_BATCH=512 timeout -k 10 900 bash tests/micro/ab_b64.sh "opt:persistent_gemm=0" > gpurun_out/r4_ab_b512_persist2.txt 2>&1
cat gpurun_out/r4_ab_b512_persist2.txt
for cfg in "burst 13 120" "burst 0 120"; do set -- $cfg
  timeout -k 10 600 python3 tests/server_load.py --streams 64 --seconds $3 --right-context $2 --mode $1 --warmup-seconds 6 --prewarm > gpurun_out/r4_server_load_$1_R$2_b.json 2> gpurun_out/r4_server_load_$1_R$2_b.err
  echo load $cfg rc $?
  cut -c1-400 gpurun_out/r4_server_load_$1_R$2_b.json
done
timeout -k 10 1100 python3 tests/micro/margin_sweep.py gpu gpurun_out/margin > gpurun_out/r4_margin_gpu.log 2>&1
echo margin rc $?
tail -3 gpurun_out/r4_margin_gpu.log | cut -c1-600
