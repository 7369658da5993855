set -x
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "post_rows" > gpurun_out/r4_post_rows_test.txt 2>&1
echo test rc $?
tail -5 gpurun_out/r4_post_rows_test.txt
timeout -k 10 700 bash tests/micro/ab_b64.sh "opt:post_rows=0" > gpurun_out/r4_ab_post_rows.txt 2>&1
cat gpurun_out/r4_ab_post_rows.txt
