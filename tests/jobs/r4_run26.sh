set -x
mkdir -p gpurun_out
cd tests/micro && timeout -k 10 300 ./persist_probe cold 896 1792 > ../../gpurun_out/r4_touch_off.txt 2>&1 && timeout -k 10 300 ./persist_probe cold pf 896 1792 > ../../gpurun_out/r4_touch_on.txt 2>&1 && timeout -k 10 300 ./persist_probe cold 896 1792 > ../../gpurun_out/r4_touch_off2.txt 2>&1 && timeout -k 10 300 ./persist_probe cold pf 896 1792 > ../../gpurun_out/r4_touch_on2.txt 2>&1; cd ../..
cat gpurun_out/r4_touch_off.txt gpurun_out/r4_touch_on.txt gpurun_out/r4_touch_off2.txt gpurun_out/r4_touch_on2.txt | cut -c1-230
