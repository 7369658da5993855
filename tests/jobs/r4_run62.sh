set -x
mkdir -p gpurun_out
AB_BATCH=128 timeout -k 10 600 bash tests/micro/ab_b64.sh "opt:probe_n1024_splits=2" > gpurun_out/r4_ab_b128_n1024b.txt 2>&1; cat gpurun_out/r4_ab_b128_n1024b.txt
AB_BATCH=112 timeout -k 10 600 bash tests/micro/ab_b64.sh "opt:probe_n1024_splits=2" > gpurun_out/r4_ab_b112_n1024b.txt 2>&1; cat gpurun_out/r4_ab_b112_n1024b.txt
