set -x
mkdir -p gpurun_out
cd tests/micro && timeout -k 10 400 ./persist_probe cold 1792 3584 7168 > ../../gpurun_out/r4_tile_order_cold.txt 2>&1; timeout -k 10 400 ./persist_probe 7168 15360 > ../../gpurun_out/r4_tile_order_hot.txt 2>&1; cd ../..
awk -F'|' '{print substr($1,1,52), "|", $2, "|", $3, "|", $4, "|", $5}' gpurun_out/r4_tile_order_cold.txt gpurun_out/r4_tile_order_hot.txt | cut -c1-260
AB_BATCH=512 timeout -k 10 500 bash tests/micro/ab_b64.sh > gpurun_out/r4_ab_b512_order.txt 2>&1
cat gpurun_out/r4_ab_b512_order.txt
