set -x
mkdir -p gpurun_out
AB_SYNC=0 AB_BATCH=128 timeout -k 10 600 bash tests/micro/ab_b64.sh "opt:probe_n1024_splits=1" "opt:probe_n1024_splits=2" > gpurun_out/r4_ab_b128_n1024.txt 2>&1; cat gpurun_out/r4_ab_b128_n1024.txt
AB_SYNC=0 AB_BATCH=96 timeout -k 10 600 bash tests/micro/ab_b64.sh "opt:probe_n1024_splits=1" "opt:probe_n1024_splits=2" > gpurun_out/r4_ab_b96_n1024.txt 2>&1; cat gpurun_out/r4_ab_b96_n1024.txt
