set -x
mkdir -p gpurun_out
cd tests/micro && timeout -k 10 400 ./persist_probe cold 3584 7168 > ../../gpurun_out/r4_wide_order_cold.txt 2>&1; timeout -k 10 400 ./persist_probe 7168 15360 > ../../gpurun_out/r4_wide_order_hot.txt 2>&1; cd ../..
awk -F'|' '{print substr($1,1,52), "|", $4, "|", $5}' gpurun_out/r4_wide_order_cold.txt gpurun_out/r4_wide_order_hot.txt
