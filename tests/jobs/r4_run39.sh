set -x
mkdir -p gpurun_out
timeout -k 10 900 python3 bench.py > gpurun_out/r4_bench_final2.json 2> gpurun_out/r4_bench_final2.err
echo bench rc $?
tail -c 3200 gpurun_out/r4_bench_final2.json
PROF_STEPS=16 timeout -k 10 700 bash tests/prof_r4.sh b64_R13 --batch 64 --right-context 13 --weights q8_0 > gpurun_out/r4_prof_b64_R13.txt 2>&1
tail -22 gpurun_out/r4_prof_b64_R13.txt
PROF_MODE="" PROF_COUNTERS="" PROF_STEPS=16 timeout -k 10 400 bash tests/prof_r4.sh b64_R13_pipelined --batch 64 --right-context 13 --weights q8_0 > gpurun_out/r4_prof_b64_R13_pipelined.txt 2>&1
tail -12 gpurun_out/r4_prof_b64_R13_pipelined.txt
